"""Latent ODE (reference: caspr/models/latent_ode_model.py) with a fixed-step RK4 HIP kernel.

Same parameter tree, including the reference's double registration of the dynamics net under
`ode_func.*` and `solver.ode_func.*` (latent_ode_model.py:36-38,81) and the `_num_evals` buffer.
DEVIATION (documented): the reference integrates with adaptive dopri5 at rtol=atol=1e-3
(latent_ode_model.py:38,83); this build runs `steps` RK4 steps per requested interval.
"""
import torch
import torch.nn as nn

from .. import ops
from ..utils.weight_cache import WeightCache


class LatentODE(nn.Module):
    def __init__(self, input_size=1024, hidden_size=1024, num_layers=2, nonlinearity=nn.Tanh, augment_size=0, rk4_steps=2):
        super(LatentODE, self).__init__()
        if nonlinearity is not nn.Tanh or num_layers != 2:
            raise ValueError("the latent RK4 kernel implements the reference configuration: 2 hidden layers, Tanh (caspr.py:61-64)")
        if augment_size != 0:
            raise ValueError("augment_size > 0 is not used by the reference model and not supported")
        self.input_size = input_size
        self.augment_size = augment_size
        self.output_size = input_size + self.augment_size
        self.rk4_steps = rk4_steps
        self.ode_func = DynamicsNet(input_size=self.output_size, hidden_size=hidden_size, num_layers=num_layers, nonlinearity=nonlinearity)
        self.solver = ODESolver(self.ode_func, method='dopri5', rtol=1e-3, atol=1e-4)
        init_network_weights(self.ode_func)
        self._cache = WeightCache()

    def get_output_size(self):
        return self.output_size

    def _weights(self):
        lin = [self.ode_func.dynamics_net[i] for i in (0, 2, 4, 6)]

        def build():
            out = []
            for l in lin:
                out += [ops.PackedWeight(l.weight.detach().contiguous()), l.bias.detach().contiguous()]
            return out
        return self._cache.get("w", [l.weight for l in lin] + [l.bias for l in lin], build)

    def forward(self, z0, t):
        """z0 (B,H) [may be a column slice of a wider (B,*) tensor], t (T,) ascending -> (B,T,H)  (latent_ode_model.py:45-70)."""
        if not z0.is_cuda:
            raise ValueError("caspr_amd.LatentODE runs on the GPU only (HIP kernels)")
        self.ode_func._num_evals.fill_(0)
        if z0.shape[1] != self.input_size:
            raise ValueError("expected %d latent dims, got %d" % (self.input_size, z0.shape[1]))
        if torch.is_grad_enabled() and (z0.requires_grad or any(p.requires_grad for p in self.ode_func.parameters())):
            from ..train.flow_grad import latent_solve_train                                    # differentiable RK4 (training)
            return latent_solve_train(self, z0, t)
        # z0 may be the view z[:, :H] of the (B,1600) encoder output: the kernel takes its row stride
        out = ops.latent_rk4(z0, t.detach().float().contiguous(), self.rk4_steps, self._weights())
        Tu = t.shape[0]
        self.ode_func._num_evals += 4 * self.rk4_steps * max(Tu - 1, 0)
        return out

    def plan_times(self, time_tensor):
        """Everything of solve_at that depends on the time stamps only (the sort, its inverse, the gather rows, the number of
        evaluations): a dozen tiny device kernels that `reconstruct` issues BEFORE the encoder, off the path between the
        encoder's last layer and the solve (they took 0.25 ms there, queued behind the T-NOCS layer's workgroups)."""
        B, T = time_tensor.shape
        flat = time_tensor.reshape(-1).float()
        sorted_t, perm = torch.sort(flat, stable=True)
        pos = torch.empty_like(perm)
        pos[perm] = torch.arange(perm.numel(), device=perm.device)
        distinct = (sorted_t[1:] != sorted_t[:-1]).sum()
        rows = torch.arange(B, device=time_tensor.device).view(-1, 1).expand(B, T)
        return {"shape": (B, T), "sorted_t": sorted_t.contiguous(), "rows": rows, "pos": pos.view(B, T),
                "evals": (4.0 * self.rk4_steps * distinct).to(self.ode_func._num_evals.dtype)}

    def solve_at(self, z0, time_tensor, plan=None):
        """z(t) for every entry of time_tensor (B,T) [any order, repeats allowed] -> (B,T,H), without a host
        synchronisation: the reference takes torch.unique of the times (caspr.py:166), whose output size is
        data-dependent; here ALL B*T stamps are sorted on the device and handed to the kernel, which skips the
        zero-length intervals between repeated stamps -- the same integration steps, the same values.
        plan: plan_times(time_tensor) made earlier on this stream."""
        if plan is None or plan["shape"] != tuple(time_tensor.shape):
            plan = self.plan_times(time_tensor)
        out = ops.latent_rk4(z0, plan["sorted_t"], self.rk4_steps, self._weights())            # (B, B*T, H)
        self.ode_func._num_evals.copy_(plan["evals"])                                           # evaluations actually run
        return out[plan["rows"], plan["pos"], :]

    def num_evals(self):
        return self.ode_func._num_evals.item()


class ODESolver(nn.Module):
    """Kept for the checkpoint surface (`solver.ode_func.*` aliases, latent_ode_model.py:76-99)."""

    def __init__(self, ode_func, method='dopri5', rtol=1e-4, atol=1e-5):
        super(ODESolver, self).__init__()
        self.method = method
        self.ode_func = ode_func
        self.rtol = rtol
        self.atol = rtol   # sic: the reference assigns rtol (latent_ode_model.py:83)
        if not isinstance(self.ode_func, nn.Module):
            raise ValueError('ode_func is required to be an instance of nn.Module to use the adjoint method')

    def forward(self, z0, t):
        raise NotImplementedError("the adaptive solver is replaced by LatentODE's RK4 kernel; call LatentODE.forward")


class DynamicsNet(nn.Module):
    """Parameter container of the dynamics MLP (latent_ode_model.py:102-147)."""

    def __init__(self, input_size=1024, hidden_size=1024, num_layers=2, nonlinearity=nn.Tanh):
        super(DynamicsNet, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.nonlinearity = nonlinearity
        self.output_size = self.input_size
        layers = [nn.Linear(self.input_size, self.hidden_size)]
        for i in range(self.num_layers):
            layers.append(self.nonlinearity())
            layers.append(nn.Linear(self.hidden_size, self.hidden_size))
        layers.append(self.nonlinearity())
        layers.append(nn.Linear(self.hidden_size, self.output_size))
        self.dynamics_net = nn.Sequential(*layers)
        self.register_buffer("_num_evals", torch.tensor(0.))

    def num_evals(self):
        return self._num_evals


def init_network_weights(net, std=0.1):
    for m in net.modules():
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, mean=0, std=std)
            nn.init.constant_(m.bias, val=0)
