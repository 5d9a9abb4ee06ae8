"""One configuration object for everything the Python host selects at run time (kernel families, stream schedules).

Rounds 2-4 grew ten environment reads at import time inside the product path (CASPR_MATMUL, CASPR_CONV_X6W, CASPR_X6W_MIN_CIN,
CASPR_LATENT_TEAM, CASPR_EARLY_LATENT, CASPR_CNF_NODE, ...): a stray variable in a user's shell silently changed kernels and speed.
Now the defaults below ARE the configuration; the environment is honoured only under CASPR_DEBUG=1 (A/B timing, debugging), and a
knob found in the environment WITHOUT CASPR_DEBUG=1 is reported once and ignored.  `active()` is what bench.py echoes in its line
(config.matrix_products.selection), read from the live module state (tests and tools may flip the module attributes at run time:
ops.set_matmul_mode, ops.CONV_X6W, ...).  The C library never reads the environment (include/caspr_hip.h)."""
import os
import warnings
from dataclasses import dataclass, asdict


@dataclass
class KernelConfig:
    matmul: str = "bf16x6"              # "bf16x6" | "f32": the matrix-product kernels (ops.set_matmul_mode switches at run time)
    conv_x6w: bool = True               # layers with >= 512 output channels on the persistent 512-channel conv (gemm_bf16x6w.hip)
    x6w_min_cin: int = 512              # ... from this many input channels
    latent_team: bool = True            # the 32-workgroup latent-ODE kernel (False: the single-workgroup kernel)
    early_latent: bool = True           # the latent solve beside the encoder's last layer (False: in front of the flow, rounds 1-3)
    early_latent_team: bool = True      # ... on the team kernel with 32 reserved compute units (False: single workgroup, one unit)
    sa_lo_parts: bool = True            # the first set-abstraction level hands hi + lo to the second (models/pointnet2.py)
    sa_scale_streams: bool = True       # the two scales of a set-abstraction level on two streams
    global_stream: bool = True          # the global PointNet on a stream of its own (models/tpointnet2.py): -0.05 .. -0.17 ms, bit-identical
    sa_pre_aggregate: bool = True       # set abstraction, wide levels: the first layer's feature part once per source point, not per (centre, sample)
    fp_commute: bool = True             # feature propagation's first conv on the coarse level (finest level: interpolation and conv commute)
    sa_f64_streams: bool = False        # the f64 re-evaluation of a scale's small balls beside its MFMA kernel, on a stream of its own (measured: + 0.4 ms, off)
    train_cnf_out_node: bool = True     # training: the ODE function's output epilogue as one node (train/flow_grad.py)
    train_cnf_hidden_node: bool = True  # training: the hidden layers with the activation backward in the data-gradient conv
    train_latent_node: bool = True      # training: the latent solve as one autograd node (team tape + team adjoint)
    train_cnf_checkpoint: bool = False  # training: keep the CNF's state per RK4 step only, recompute the step in the backward pass
                                        # (63 GB -> ~1/8 of tape at the cfg-3 shard for one more forward of the block)


# environment name -> (field, parser); read only under CASPR_DEBUG=1
_ENV = {
    "CASPR_MATMUL": ("matmul", lambda v: v.strip().lower()),
    "CASPR_CONV_X6W": ("conv_x6w", lambda v: v != "0"),
    "CASPR_X6W_MIN_CIN": ("x6w_min_cin", int),
    "CASPR_LATENT_TEAM": ("latent_team", lambda v: v != "0"),
    "CASPR_EARLY_LATENT": (None, None),            # "0": early_latent off; "single": early_latent_team off
    "CASPR_SA_LO_PARTS": ("sa_lo_parts", lambda v: v != "0"),
    "CASPR_SA_SCALE_STREAMS": ("sa_scale_streams", lambda v: v != "0"),
    "CASPR_SA_F64_STREAMS": ("sa_f64_streams", lambda v: v != "0"),
    "CASPR_FP_COMMUTE": ("fp_commute", lambda v: v != "0"),
    "CASPR_SA_PRE_AGGREGATE": ("sa_pre_aggregate", lambda v: v != "0"),
    "CASPR_GLOBAL_STREAM": ("global_stream", lambda v: v != "0"),
    "CASPR_CNF_OUT_NODE": ("train_cnf_out_node", lambda v: v != "0"),
    "CASPR_CNF_NODE": ("train_cnf_hidden_node", lambda v: v != "0"),
    "CASPR_LATENT_NODE": ("train_latent_node", lambda v: v != "0"),
    "CASPR_CNF_CHECKPOINT": ("train_cnf_checkpoint", lambda v: v != "0"),
}


def load(environ=None):
    """-> KernelConfig: the defaults, overridden from the environment only when CASPR_DEBUG=1."""
    env = os.environ if environ is None else environ
    cfg = KernelConfig()
    present = [k for k in _ENV if k in env]
    if env.get("CASPR_DEBUG", "0") != "1":
        if present:
            warnings.warn("caspr_amd: %s found in the environment but CASPR_DEBUG=1 is not set -- ignored (the kernel selection is "
                          "caspr_amd.config.config; debugging knobs need CASPR_DEBUG=1)" % ", ".join(sorted(present)), RuntimeWarning, stacklevel=2)
        return cfg
    for name in present:
        field, parse = _ENV[name]
        if name == "CASPR_EARLY_LATENT":
            cfg.early_latent = env[name] != "0"
            cfg.early_latent_team = env[name] != "single"
        else:
            setattr(cfg, field, parse(env[name]))
    if cfg.matmul not in ("bf16x6", "f32"):
        raise ValueError("CASPR_MATMUL must be 'bf16x6' or 'f32', got %r" % cfg.matmul)
    return cfg


config = load()


def active():
    """The selection in force NOW (module state, which tests / tools / bench.py may have switched since import)."""
    from . import ops
    from .models import caspr as _c, pointnet2 as _p, tpointnet2 as _t
    d = asdict(config)
    d.update({"matmul": ops.matmul_mode(), "conv_x6w": ops.CONV_X6W, "x6w_min_cin": ops._X6W_MIN_CIN, "latent_team": ops.LATENT_TEAM,
              "early_latent": _c.EARLY_LATENT, "early_latent_team": _c.EARLY_LATENT_TEAM, "sa_lo_parts": _p.LO_PARTS,
              "sa_scale_streams": _p.SCALE_STREAMS, "sa_f64_streams": _p.F64_STREAMS, "fp_commute": _p.FP_COMMUTE, "sa_pre_aggregate": _p.PRE_AGGREGATE, "global_stream": _t.GLOBAL_STREAM, "debug_env": os.environ.get("CASPR_DEBUG", "0") == "1"})
    return d
