"""CNF block and SequentialFlow (reference: caspr/models/cnf.py) on the fused RK4 HIP kernel.

DEVIATION (documented): the reference integrates with adaptive dopri5 at atol=rtol=1e-5
through torchdiffeq (cnf.py:100-119, flow.py:96-99); this build runs `rk4_steps` classic RK4 steps
over [0, sqrt_end_time^2] inside ONE kernel launch.  When no log-density is requested (sampling,
cnf.py:71-74 with logpx=None) the Hutchinson divergence is skipped: with a fixed step the xyz
trajectory does not depend on it.
"""
import torch
import torch.nn as nn

from .. import ops
from ..utils.weight_cache import WeightCache
from .normalization import MovingBatchNorm1d

__all__ = ["CNF", "SequentialFlow"]


class CNF(nn.Module):
    def __init__(self, odefunc, conditional=True, T=1.0, train_T=False, solver='dopri5', atol=1e-5, rtol=1e-5,
                 use_adjoint=True, rk4_steps=8):
        super(CNF, self).__init__()
        self.train_T = train_T
        self.T = T
        if train_T:
            self.register_parameter("sqrt_end_time", nn.Parameter(torch.sqrt(torch.tensor(T))))
        self.use_adjoint = use_adjoint
        self.odefunc = odefunc
        self.solver = solver
        self.atol = atol
        self.rtol = rtol
        self.test_solver = solver
        self.test_atol = atol
        self.test_rtol = rtol
        self.solver_options = {}
        self.conditional = conditional
        self.rk4_steps = rk4_steps
        self._count_evals = True      # False while the accuracy guard repeats a solve (CaSPR._guard_cnf): get_nfe() counts the real one only
        self._narrow = False          # True while the guard runs its check solve: the 64-point sampling kernel (ops.cnf_rk4(narrow=True))
        self._cache = WeightCache()

    def _weights(self):
        layers = self.odefunc.diffeq.layers
        srcs = []
        for l in layers:
            srcs += [l._layer.weight, l._layer.bias, l._hyper_bias.weight, l._hyper_gate.weight, l._hyper_gate.bias]

        def build():
            H = layers[0]._layer.out_features
            gates = torch.cat([l._hyper_gate.weight.detach() for l in layers], dim=0)           # (3H+3, 1+zdim)
            biases = torch.cat([l._hyper_bias.weight.detach() for l in layers], dim=0)
            whyp = torch.cat([gates, biases], dim=0).contiguous()                               # rows [gate l0..l3 | bias l0..l3]
            gate_b = torch.cat([l._hyper_gate.bias.detach() for l in layers])
            return {
                "H": H,
                "hyp": ops.PackedWeight(whyp, col0=1),                                           # columns 1.. multiply the context (odefunc.py:133)
                "hyp_bias": torch.cat([gate_b, torch.zeros_like(gate_b)]).contiguous(),
                "tcol": whyp[:, 0].contiguous(),                                                # column 0 multiplies t
                "w0": layers[0]._layer.weight.detach().contiguous(), "b0": layers[0]._layer.bias.detach().contiguous(),
                "w1p": ops.PackedWeight(layers[1]._layer.weight.detach().contiguous()), "b1": layers[1]._layer.bias.detach().contiguous(),
                "w2p": ops.PackedWeight(layers[2]._layer.weight.detach().contiguous()), "b2": layers[2]._layer.bias.detach().contiguous(),
                "w3": layers[3]._layer.weight.detach().contiguous(), "b3": layers[3]._layer.bias.detach().contiguous(),
            }
        return self._cache.get("w", srcs, build)

    def _weights_x6(self):
        """Three-plane bf16 packs of the two hidden layers for the bf16x6 kernel (built on first use)."""
        layers = self.odefunc.diffeq.layers
        if layers[0]._layer.out_features != 512:
            return None, None
        return self._cache.get("wx6", [layers[1]._layer.weight, layers[2]._layer.weight],
                               lambda: (ops.pack_cnf_x6(layers[1]._layer.weight.detach().contiguous()),
                                        ops.pack_cnf_x6(layers[2]._layer.weight.detach().contiguous())))

    def end_time(self):
        """T_end = sqrt_end_time^2 (cnf.py:87-90), read back once per parameter version (a D2H sync otherwise
        sits between the encoder and every CNF launch)."""
        if self.train_T:
            p = self.sqrt_end_time
            return self._cache.get("t_end", [p], lambda: float(p.detach() * p.detach()))
        return float(self.T)

    def integrate(self, x, context, logpx, reverse, mbn_in=None, mbn_out=None):
        """x (BT,n,3), context (BT,zdim), logpx (BT,n,1)|None.  MBN params fused at either end (kernel_params())."""
        if not x.is_cuda:
            raise ValueError("caspr_amd.CNF runs on the GPU only (HIP kernels)")
        if not self.conditional:
            raise ValueError("only the conditional CNF (flow.py:78-81) is supported")
        assert context is not None                                                              # cnf.py:78
        w = self._weights()
        # frames are the ROWS of this conv: row-invariant, so a frame's gates do not depend on the batch around it
        hyper = ops.conv1x1(w["hyp"], w["hyp_bias"], context.contiguous().view(1, context.shape[0], -1), row_invariant=True)[0]   # (BT, 2*(3H+3) padded)
        e = None
        if logpx is not None:
            e = self.odefunc._e
            if e is None:
                e = torch.randn_like(x)                                                         # odefunc.py:127-128
            self.odefunc._e = e
        w1x, w2x = self._weights_x6() if ops.CNF_BF16X6 else (None, None)
        if ops.BEFORE_CNF_LAUNCH is not None and self._count_evals:      # (not for the accuracy guard's check solve)
            hook, ops.BEFORE_CNF_LAUNCH = ops.BEFORE_CNF_LAUNCH, None
            hook()
        res = ops.cnf_rk4(x.contiguous(), hyper, w["tcol"], w["w0"], w["b0"], w["w1p"], w["b1"], w["w2p"], w["b2"], w["w3"], w["b3"],
                          self.end_time(), self.rk4_steps, reverse, mbn_in, mbn_out, e=e,
                          logp=None if logpx is None else logpx.contiguous(), w1x=w1x, w2x=w2x, narrow=self._narrow)
        if self._count_evals:
            self.odefunc._num_evals += 4 * self.rk4_steps
        return res

    def forward(self, x, context=None, logpx=None, integration_times=None, reverse=False):
        """Reference signature (cnf.py:70).  integration_times must be None (the learned end time is used)."""
        if integration_times is not None:
            raise ValueError("custom integration_times are not supported by the fused RK4 kernel")
        self.odefunc.before_odeint()        # cnf.py:100: clears the Hutchinson noise -> a fresh draw per solve (odefunc.py:127-128)
        return self.integrate(x, context, logpx, reverse)

    def num_evals(self):
        return self.odefunc._num_evals.item()


class SequentialFlow(nn.Module):
    """Container [MBN, CNF x k, MBN] (cnf.py:20-48, flow.py:68-72).  The first / last MovingBatchNorm in
    the direction of travel are fused into the prologue / epilogue of the adjacent CNF launch."""

    def __init__(self, layer_list, use_bn=True):
        super(SequentialFlow, self).__init__()
        self.chain = nn.ModuleList(layer_list)
        self.use_bn = use_bn

    def forward(self, x, context, logpx=None, reverse=False, inds=None, integration_times=None, e=None):
        if integration_times is not None:
            raise ValueError("custom integration_times are not supported by the fused RK4 kernel")
        if inds is None:
            inds = list(range(len(self.chain) - 1, -1, -1)) if reverse else list(range(len(self.chain)))
        inds = list(inds)
        shape = x.shape
        x = x.reshape(-1, shape[-2], shape[-1]) if x.dim() != 3 else x
        k = 0
        while k < len(inds):
            layer = self.chain[inds[k]]
            if isinstance(layer, MovingBatchNorm1d):
                nxt = self.chain[inds[k + 1]] if k + 1 < len(inds) else None
                if isinstance(nxt, CNF):
                    # fuse: MBN -> CNF [-> MBN]
                    mbn_in = layer.kernel_params()                                              # pre-update statistics (normalization.py:60-61)
                    if self.training and not reverse:
                        layer.update_running_mean(x)                                            # normalization.py:63-64
                    after = self.chain[inds[k + 2]] if k + 2 < len(inds) else None
                    fuse_out = isinstance(after, MovingBatchNorm1d) and not (self.training and not reverse)
                    nxt.odefunc.before_odeint(e)
                    res = nxt.integrate(x, context, logpx, reverse, mbn_in=mbn_in, mbn_out=after.kernel_params() if fuse_out else None)
                    x, logpx = (res, None) if logpx is None else res
                    k += 3 if fuse_out else 2
                    continue
                res = layer(x, context, logpx, integration_times, reverse)
            else:
                if isinstance(layer, CNF):
                    layer.odefunc.before_odeint(e)
                    res = layer.integrate(x, context, logpx, reverse)
                else:
                    res = layer(x, context, logpx, integration_times, reverse)
            x, logpx = (res, None) if logpx is None else res
            k += 1
        return x if logpx is None else (x, logpx)
