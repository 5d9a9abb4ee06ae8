"""Point-CNF construction (reference: caspr/models/flow.py)."""
from .odefunc import ODEfunc, ODEnet
from .normalization import MovingBatchNorm1d
from .cnf import CNF, SequentialFlow
from .latent_ode_model import LatentODE


def count_nfe(model):
    class AccNumEvals(object):
        def __init__(self):
            self.num_evals = 0

        def __call__(self, module):
            if isinstance(module, CNF) or isinstance(module, LatentODE):
                self.num_evals += module.num_evals()

    accumulator = AccNumEvals()
    model.apply(accumulator)
    return accumulator.num_evals


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def build_model(args, input_dim, hidden_dims, context_dim, num_blocks, conditional):
    def build_cnf():
        diffeq = ODEnet(hidden_dims=hidden_dims, input_shape=(input_dim,), context_dim=context_dim,
                        layer_type=args.layer_type, nonlinearity=args.nonlinearity)
        odefunc = ODEfunc(diffeq=diffeq)
        return CNF(odefunc=odefunc, T=args.time_length, train_T=args.train_T, conditional=conditional, solver=args.solver,
                   use_adjoint=args.use_adjoint, atol=args.atol, rtol=args.rtol, rk4_steps=args.rk4_steps)

    chain = [build_cnf() for _ in range(num_blocks)]
    if args.batch_norm:
        chain = [MovingBatchNorm1d(input_dim)] + chain + [MovingBatchNorm1d(input_dim)]
    return SequentialFlow(chain, use_bn=args.batch_norm)


def get_point_cnf(args):
    """flow.py:78-83 without the unconditional `.cuda()`: the caller moves the model (model.to(device))."""
    dims = tuple(map(int, args.dims.split("-")))
    if dims != (512, 512, 512) or args.input_dim != 3:
        raise ValueError("the CNF kernel is built for input_dim=3 and dims 512-512-512 (flow.py:88-89)")
    return build_model(args, args.input_dim, dims, args.zdim, args.num_blocks, True)


class PointCNFArgs():
    """flow.py:86-100 defaults, plus the fixed-step count of this build."""

    def __init__(self):
        self.input_dim = 3
        self.dims = "512-512-512"
        self.zdim = 512
        self.num_blocks = 1
        self.layer_type = 'concatsquash'
        self.nonlinearity = 'softplus'
        self.time_length = 0.5
        self.train_T = True
        self.solver = 'dopri5'
        self.use_adjoint = True
        self.atol = 1e-5
        self.rtol = 1e-5
        self.batch_norm = True
        self.rk4_steps = 8
