"""ctypes binding of libcaspr_hip.so (include/caspr_hip.h, include/caspr_hip_train.h).

The library is built in-tree by caspr_amd/csrc/build.py (hipcc --offload-arch=gfx950).  There is no
CPU fallback: if the shared object is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libcaspr_hip.so")

c_fp = ctypes.c_void_p   # const float* / float* (device)
c_ip = ctypes.c_void_p   # const int32_t* / int32_t* (device)
c_int = ctypes.c_int
c_long = ctypes.c_long
c_float = ctypes.c_float
c_stream = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/caspr_hip.h one to one
SIGNATURES = {
    "caspr_last_error_string": (ctypes.c_char_p, []),
    "caspr_abi_version": (c_int, []),
    "caspr_prep_input_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_stream]),
    "caspr_fps_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_ip, c_fp, c_stream]),
    "caspr_gather_points_f32": (c_int, [c_fp, c_int, c_ip, c_int, c_int, c_int, c_int, c_fp, c_int, c_stream]),
    "caspr_ball_query_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_float, c_int, c_ip, c_stream]),
    "caspr_ball_query2_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_float, c_int, c_ip, c_float, c_int, c_ip, c_stream]),
    "caspr_group_points_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_ip, c_int, c_int, c_int, c_int, c_int, c_fp, c_stream]),
    "caspr_sa_mlp_max_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_ip, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_fp, c_fp, c_fp, c_fp, c_int,
                                     c_fp, c_fp, c_fp, c_fp, c_int,
                                     c_fp, c_fp, c_fp, c_fp, c_int,
                                     c_fp, c_int, c_int, c_stream]),
    "caspr_sa_mlp_max_workspace_ints": (c_long, [c_int, c_int]),
    "caspr_sa_mlp_max_ws_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_ip, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_fp, c_fp, c_fp, c_fp, c_int,
                                        c_fp, c_fp, c_fp, c_fp, c_int,
                                        c_fp, c_fp, c_fp, c_fp, c_int,
                                        c_fp, c_int, c_int, c_ip, c_stream]),
    "caspr_sa_mlp_max_pre_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_ip, c_int, c_int, c_int, c_int,
                                         c_fp, c_fp, c_fp, c_fp, c_int,
                                         c_fp, c_fp, c_fp, c_fp, c_int,
                                         c_fp, c_fp, c_fp, c_fp, c_int,
                                         c_fp, c_int, c_int, c_stream]),
    "caspr_group_rows_pre_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_ip, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_stream]),
    "caspr_three_nn_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_ip, c_fp, c_stream]),
    "caspr_three_interp_f32": (c_int, [c_fp, c_int, c_ip, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_int, c_stream]),
    "caspr_three_interp_add_gn_ws_bytes": (c_long, [c_int, c_int, c_int]),
    "caspr_three_interp_add_gn_f32": (c_int, [c_fp, c_int, c_ip, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int,
                                              c_int, c_fp, c_fp, c_float, c_fp, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_packed_size": (c_long, [c_int, c_int]),
    "caspr_pack_weight_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_stream]),
    "caspr_conv1x1_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_stream]),
    "caspr_bf16x3_packed_bytes": (c_long, [c_int, c_int]),
    "caspr_pack_weight_bf16x3": (c_int, [c_fp, c_int, c_int, c_int, c_int, ctypes.c_void_p, c_stream]),
    "caspr_conv1x1_bf16x6_f32": (c_int, [ctypes.c_void_p, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_stream]),
    "caspr_conv_gn_ws_bytes": (c_long, [c_int, c_int, c_int]),
    "caspr_conv1x1_gn_bf16x6_f32": (c_int, [ctypes.c_void_p, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_fp, c_fp, c_float, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_conv1x1_gn_pooled_bf16x6_f32": (c_int, [ctypes.c_void_p, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_int, c_int,
                                                   c_int, c_int, c_fp, c_fp, c_float, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_x6w_packed_bytes": (c_long, [c_int, c_int]),
    "caspr_pack_weight_x6w": (c_int, [c_fp, c_int, c_int, c_int, c_int, ctypes.c_void_p, c_stream]),
    "caspr_conv1x1_x6w_f32": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_fp, c_fp, c_float, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_conv1x1_x6w_pooled_f32": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_int,
                                             c_int, c_int, c_int, c_fp, c_fp, c_float, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_conv1x1_x6w_part_f32": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_fp, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_int, c_int, ctypes.c_void_p, c_long, c_stream]),
    "caspr_conv_gn_finalize_f32": (c_int, [ctypes.c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_float, c_fp, c_fp, c_fp, c_fp,
                                           c_fp, c_stream]),
    "caspr_gn_ws_bytes": (c_long, [c_int, c_int, c_int, c_int]),
    "caspr_gn_stats_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_float, c_fp, c_fp, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_latent_rk4_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_stream]),
    "caspr_latent_team_ws_bytes": (c_long, [c_int]),
    "caspr_latent_rk4_team_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                          ctypes.c_void_p, c_long, c_stream]),
    "caspr_latent_rk4_team_tape_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                               c_fp, c_int, c_fp, c_fp, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_latent_rk4_team_adjoint_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                                  c_fp, c_fp, c_int, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_cnf_rk4_f32": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_float, c_int, c_int,
                                  c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_stream]),
    "caspr_cnf_x6_packed_bytes": (c_long, []),
    "caspr_pack_weight_cnf_x6": (c_int, [c_fp, c_int, ctypes.c_void_p, c_stream]),
    "caspr_cnf_rk4_x6_f32": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp, ctypes.c_void_p, c_fp, ctypes.c_void_p, c_fp, c_fp, c_fp, c_int, c_float,
                                     c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_stream]),
    "caspr_chamfer_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_stream]),
    "caspr_emd_ws_bytes": (c_long, [c_int, c_int, c_int]),
    "caspr_emd_f32": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, ctypes.c_void_p, c_long, c_stream]),
    # ---- include/caspr_hip_train.h (training tier) ----
    "caspr_gn_stats_train_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_float, c_fp, c_fp, c_fp, c_fp, c_fp,
                                         ctypes.c_void_p, c_long, c_stream]),
    "caspr_wgrad_ws_bytes": (c_long, [c_long, c_int, c_int]),
    "caspr_conv1x1_wgrad_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_int,
                                        ctypes.c_void_p, c_long, c_stream]),
    "caspr_conv1x1_wgrad_bf16x6_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_int,
                                        ctypes.c_void_p, c_long, c_stream]),
    "caspr_gn_bwd_ws_bytes": (c_long, [c_long, c_int, c_int, c_int]),
    "caspr_gn_bwd_f32": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_ip, c_fp, c_int, c_long, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp,
                                 c_int, c_fp, c_fp, c_int, ctypes.c_void_p, c_long, c_stream]),
    "caspr_argmax_ws_bytes": (c_long, [c_long, c_int, c_int]),
    "caspr_argmax_points_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_ip, ctypes.c_void_p, c_long, c_stream]),
    "caspr_colsum_ws_bytes": (c_long, [c_long, c_int, c_int]),
    "caspr_colsum_batched_f32": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, ctypes.c_void_p, c_long, c_stream]),
    "caspr_three_interp_bwd_f32": (c_int, [c_fp, c_int, c_ip, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_stream]),
    "caspr_group_rows_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_ip, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_int, c_stream]),
    "caspr_group_rows_bwd_f32": (c_int, [c_fp, c_int, c_ip, c_int, c_int, c_int, c_int, c_int, c_fp, c_int, c_stream]),
    "caspr_segment_sum_f32": (c_int, [c_fp, c_int, c_int, c_ip, c_ip, c_fp, c_long, c_int, c_fp, c_int, c_int, c_stream]),
    "caspr_gn_rows_f32": (c_int, [c_fp, c_int, c_long, c_int, c_int, c_fp, c_fp, c_float, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_ip,
                                  c_stream]),
    "caspr_cnf_act_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_long, c_int, c_int, c_long, c_fp, c_int, c_stream]),
    "caspr_cnf_act_bwd_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_long, c_int, c_int, c_long, c_fp, c_int, c_fp, c_fp, c_stream]),
    "caspr_cnf_out_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_fp, c_long, c_int, c_long, c_fp, c_fp, c_stream]),
    "caspr_cnf_out_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_long, c_int, c_long, c_fp, c_fp, c_fp, c_stream]),
    "caspr_cnf_act_bwd_out_f32": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_long, c_int, c_int, c_long, c_fp, c_int, c_fp, c_fp, c_stream]),
    "caspr_conv1x1_cnf_act_bf16x6_f32": (c_int, [ctypes.c_void_p, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_stream]),
    "caspr_conv1x1_cnf_act_bwd_ws_bytes": (c_long, [c_int, c_int, c_int]),
    "caspr_conv1x1_cnf_act_bwd_bf16x6_f32": (c_int, [ctypes.c_void_p, c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, ctypes.c_void_p, c_long,
                                             c_int, c_int, c_int, c_int, c_stream]),
    "caspr_cnf_in_bwd_chunk": (c_int, [c_int]),
    "caspr_cnf_in_bwd_splits": (c_int, [c_int, c_int]),
    "caspr_cnf_in_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_long, c_int, c_int, c_long, c_fp, c_stream]),
    "caspr_cnf_in_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_long, c_int, c_int, c_long, c_fp, c_fp, c_fp, c_fp, c_stream]),
    "caspr_gn_rows_bwd_ws_bytes": (c_long, [c_int]),
    "caspr_gn_rows_bwd_f32": (c_int, [c_fp, c_int, c_long, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_ip,
                                      c_fp, c_int, c_fp, c_fp, c_int, ctypes.c_void_p, c_long, c_stream]),
}

_lib = None


class CasprHipError(RuntimeError):
    pass


def load():
    """Load libcaspr_hip.so and attach the C signatures.  Raises if the extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise CasprHipError(
                "libcaspr_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python caspr_amd/csrc/build.py`; there is no CPU fallback." % SO_PATH)
        lib = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().caspr_last_error_string()
        raise CasprHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))
