"""Assembly of the point CNF: [MovingBatchNorm, CNF x num_blocks, MovingBatchNorm].

Counterpart of the reference's caspr/models/flow.py (its public names -- `get_point_cnf`, `build_model`, `count_nfe`,
`count_parameters`, `PointCNFArgs` -- and the attribute names callers read, `cnf_args.zdim` / `.input_dim`
(caspr.py:101,229), are the interface; the bodies are this build's).  Differences: no unconditional `.cuda()`
(flow.py:81 -- the caller moves the model), the fixed RK4 step count is part of the options, and configurations the
fused kernel does not implement are refused instead of silently constructed.
"""
from .cnf import CNF, SequentialFlow
from .latent_ode_model import LatentODE
from .normalization import MovingBatchNorm1d
from .odefunc import ODEfunc, ODEnet


def count_nfe(model):
    """Function evaluations spent by the last solve of every integrator under `model` (caspr.py:198-202 sums the latent
    ODE and the point CNF separately through this)."""
    return sum(m.num_evals() for m in model.modules() if isinstance(m, (CNF, LatentODE)))


def count_parameters(model):
    """Number of trainable scalars."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


class PointCNFArgs:
    """Options of the point CNF with the reference's defaults (flow.py:86-100).  `rk4_steps` is this build's own knob: the
    number of fixed RK4 steps that replaces dopri5 at atol = rtol = 1e-5."""
    DEFAULTS = dict(input_dim=3, dims="512-512-512", zdim=512, num_blocks=1, layer_type="concatsquash", nonlinearity="softplus",
                    time_length=0.5, train_T=True, solver="dopri5", use_adjoint=True, atol=1e-5, rtol=1e-5, batch_norm=True,
                    rk4_steps=8)

    def __init__(self, **overrides):
        unknown = set(overrides) - set(self.DEFAULTS)
        if unknown:
            raise TypeError("unknown point-CNF option(s): %s" % ", ".join(sorted(unknown)))
        for name, value in {**self.DEFAULTS, **overrides}.items():
            setattr(self, name, value)


def _cnf_block(args, input_dim, hidden_dims, context_dim, conditional):
    net = ODEnet(hidden_dims=hidden_dims, input_shape=(input_dim,), context_dim=context_dim, layer_type=args.layer_type,
                 nonlinearity=args.nonlinearity)
    return CNF(odefunc=ODEfunc(diffeq=net), T=args.time_length, train_T=args.train_T, conditional=conditional, solver=args.solver,
               atol=args.atol, rtol=args.rtol, use_adjoint=args.use_adjoint, rk4_steps=args.rk4_steps)


def build_model(args, input_dim, hidden_dims, context_dim, num_blocks, conditional):
    """`num_blocks` CNF blocks in sequence, bracketed by MovingBatchNorm layers when args.batch_norm (checkpoint keys
    point_cnf.chain.{0 .. num_blocks+1})."""
    blocks = [_cnf_block(args, input_dim, hidden_dims, context_dim, conditional) for _ in range(num_blocks)]
    if args.batch_norm:
        blocks = [MovingBatchNorm1d(input_dim), *blocks, MovingBatchNorm1d(input_dim)]
    return SequentialFlow(blocks, use_bn=args.batch_norm)


def get_point_cnf(args):
    """The conditional flow CaSPR decodes with (caspr.py:70-72).  Stays on the CPU until the caller's model.to(device)."""
    hidden = tuple(int(d) for d in args.dims.split("-"))
    if hidden != (512, 512, 512) or args.input_dim != 3:
        raise ValueError("the CNF kernel is built for input_dim=3 and dims 512-512-512 (flow.py:88-89); got %s / %s"
                         % (args.input_dim, args.dims))
    return build_model(args, args.input_dim, hidden, args.zdim, args.num_blocks, conditional=True)
