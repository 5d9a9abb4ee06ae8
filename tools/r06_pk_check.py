#!/usr/bin/env python
"""Packed-f32 VALU arithmetic against scalar arithmetic on the same registers (tools/micro/pk_check.hip), alone and beside kernels of the encoder
on another stream.  Counts (thread, iteration) pairs whose two results differ in any bit."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
dev = torch.device("cuda:0")
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libpk_check.so"))
L.pk_check.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
torch.manual_seed(0)
pts = (torch.rand(160, 512, 3, device=dev) + torch.tensor([0.0, 0.0, 2.0], device=dev)).contiguous()
def mk(B, P, Cin, Cout):
    w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    return dict(pw=ops.PackedWeight(w), bias=torch.randn(Cout, device=dev), x=torch.randn(B, P, Cin, device=dev), sc=torch.rand(B, Cin, device=dev) + 0.5,
                sh=torch.randn(B, Cin, device=dev), g=torch.ones(Cout, device=dev), be=torch.zeros(Cout, device=dev))
LAY = {"conv128_1024 stats only": (mk(16, 20480, 128, 1024), False), "conv128_1024 with output": (mk(16, 20480, 128, 1024), True), "conv512_512": (mk(160, 1024, 512, 512), True),
       "conv1600_1600": (mk(16, 20480, 1600, 1600), True)}
side = torch.cuda.Stream()
seen_detail = []
if "--f64" in sys.argv:
    # the same question for f64 VALU arithmetic on a register pair ds_read_b64 has just returned (tools/micro/f64_check.hip): first evaluation vs a second
    # one a few instructions later
    F = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libf64_check.so"))
    F.f64_check.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for which in ["alone"] + list(LAY):
        res = []
        for r in range(6):
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                assert F.f64_check(pts.data_ptr(), 160, 512, 4000, bad.data_ptr(), side.cuda_stream) == 0
            if which != "alone":
                Ld, wr = LAY[which]
                keep = ops.conv1x1_gn(Ld["pw"], Ld["bias"], Ld["x"], Ld["g"], Ld["be"], in_scale=Ld["sc"], in_shift=Ld["sh"], in_relu=True, want_max=True, write=wr)
            torch.cuda.synchronize()
            res.append(int(bad))
        print("f64 arithmetic behind ds_read_b64, first vs repeated evaluation, %-28s: disagreements per run %s" % (which, res), flush=True)
    sys.exit(0)
for which in ["alone"] + list(LAY):
    res = []
    for r in range(6):
        bad = torch.zeros(6, dtype=torch.int32, device=dev); first = torch.zeros(1, dtype=torch.int32, device=dev); detail = torch.zeros(16, dtype=torch.int32, device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rc = L.pk_check(pts.data_ptr(), 160, 512, 4000, bad.data_ptr(), first.data_ptr(), detail.data_ptr(), side.cuda_stream)
            assert rc == 0, rc
        if which != "alone":
            Ld, wr = LAY[which]
            keep = ops.conv1x1_gn(Ld["pw"], Ld["bias"], Ld["x"], Ld["g"], Ld["be"], in_scale=Ld["sc"], in_shift=Ld["sh"], in_relu=True, want_max=True, write=wr)
        torch.cuda.synchronize()
        res.append(tuple(bad.tolist()))
        if int(detail[0]) and not seen_detail:
            seen_detail.append(1)
            d = detail.cpu()
            fl = d[4:].view(torch.float32).tolist()
            print("   one disagreement: thread %d (lane %d) iteration %d frame %d: packed (%.9g, %.9g) scalar (%.9g, %.9g) packed-again (%.9g, %.9g); point0 (%.7g, %.7g, %.7g) reference (%.7g, %.7g, %.7g)" % (
                int(d[1]), int(d[1]) % 64, int(d[2]), int(d[3]), fl[0], fl[2], fl[1], fl[3], fl[10], fl[11], fl[4], fl[5], fl[6], fl[7], fl[8], fl[9]), flush=True)
    print("%-28s: per run (packed != scalar, pk_add != sub, pk_mul != mul, pk_fma != fma, packed-again != scalar, packed-again != packed) %s" % (which, res), flush=True)
