#!/usr/bin/env python
"""HBM-side traffic of the encoder's pointwise conv layers, LAYER BY LAYER (round-5 review: one average over every launch of the
persistent conv kernel hid a piece that reads 5 GB for a 2.1 GB input).  The counters cannot tell the launches of one kernel apart by
shape (the persistent kernel's grid is the CU count), so this script runs the layers of cfg-2 one after the other, each preceded by a
MARKER launch of a kernel no conv uses (chamfer_kernel), and the join below cuts the per-dispatch counter dump at the markers.

  on the GPU box (tools/profile_round.sh):
    for C in FETCH_SIZE WRITE_SIZE: rocprofv3 --kernel-trace --pmc $C -d /tmp/pc -o r -- python tools/conv_layers_pmc.py run
        python tools/rocprof_pmc_summary.py <db> /tmp/conv_$C.txt --dispatches "conv1x1|chamfer_kernel|conv_gn"
    python tools/conv_layers_pmc.py join <fetch dump> <write dump (may be the same file)> <committed name> -> profiles/kernel_traffic.json
       entries "conv_layer:<cin>:<cout>:<rows>:<BxTxN>" = {fetch_size_kb, write_size_kb (sum over the layer's launches: main tiles + tail +
       finalize), launches, algorithmic_kb}
"""
import json
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
# cfg-2 (B = 16, T = 10, N = 2048): (batch entries, points per entry, Cin, Cout, conv -> GroupNorm?) in the order bench.py's table lists them
LAYERS = [(16, 20480, 1600, 1600, True), (16, 20480, 576, 1600, True), (160, 2048, 512, 512, True), (160, 1024, 608, 512, True),
          (16, 20480, 128, 1024, True), (160, 1024, 512, 512, True), (160, 512, 640, 512, True), (160, 512, 512, 512, True),
          (160, 256, 768, 512, True), (160, 512, 256, 512, True), (160, 256, 512, 512, True), (160, 256, 256, 512, True)]
WL = "16x10x2048"
REPS = 2


def run():
    import torch
    from caspr_amd import ops
    dev = torch.device("cuda:0")
    a = torch.rand(1, 64, 3, device=dev)
    for (B, P, Cin, Cout, gn) in LAYERS:
        w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
        bias = torch.randn(Cout, device=dev)
        x = torch.randn(B, P, Cin, device=dev)
        sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
        g, be = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        pw = ops.PackedWeight(w)
        out = torch.empty(B, P, Cout, device=dev)
        ops.conv1x1_gn(pw, bias, x, g, be, in_scale=sc, in_shift=sh, in_relu=True, out=out)      # warm-up: weight pack, workspace
        torch.cuda.synchronize()
        for _ in range(REPS):
            ops.chamfer_distance(a, a)                                                                   # the marker
            ops.conv1x1_gn(pw, bias, x, g, be, in_scale=sc, in_shift=sh, in_relu=True, out=out)
        torch.cuda.synchronize()
        del x, out, w, pw
        torch.cuda.empty_cache()
    ops.chamfer_distance(a, a)
    torch.cuda.synchronize()


def parse(path, counter):
    """-> [[(counter value, kernel), ...] per marker-delimited group], in dispatch order (the file may hold both counters' passes)"""
    groups, cur = [], None
    for ln in open(path):
        m = re.match(r"^D (\d+) (\S+) ([0-9.e+-]+) (\d+) (.*)$", ln.rstrip("\n"))
        if not m or m.group(2) != counter:
            continue
        if "chamfer_kernel" in m.group(5):          # (a marker is two launches: both directions of the distance)
            if cur:
                groups.append(cur)
            cur = []
        elif cur is not None:
            cur.append((float(m.group(3)), m.group(5)))
    return groups


def join(fetch_path, write_path, committed):
    f, w = parse(fetch_path, "FETCH_SIZE"), parse(write_path, "WRITE_SIZE")
    assert len(f) == len(w) == len(LAYERS) * REPS, (len(f), len(w), len(LAYERS) * REPS)
    path = os.path.join(ROOT, "profiles", "kernel_traffic.json")
    tab = json.load(open(path)) if os.path.exists(path) else {}
    for i, (B, P, Cin, Cout, gn) in enumerate(LAYERS):
        fk = [sum(v for v, _ in f[i * REPS + r]) for r in range(REPS)]
        wk = [sum(v for v, _ in w[i * REPS + r]) for r in range(REPS)]
        launches = [k.split("(")[0][-48:] + ":%.0f" % v for v, k in f[i * REPS]]
        rows = B * P
        tab["conv_layer:%d:%d:%d:%s" % (Cin, Cout, rows, WL)] = {
            "fetch_size_kb": sum(fk) / REPS, "write_size_kb": sum(wk) / REPS, "fetch_correction": 2.0, "launches_fetch_kb": launches,
            "algorithmic_kb": rows * (Cin + Cout) * 4 / 1024.0, "source": "profiles/%s" % committed,
            "note": "sum over ALL launches of one conv -> GroupNorm call of this shape (main channel tiles, channel remainder, statistics finalize), the layer run "
                    "on its own between marker launches (tools/conv_layers_pmc.py); gfx950: FETCH_SIZE x 2 for wide coalesced reads (MI355X_MICROARCH.md)"}
    json.dump(tab, open(path, "w"), indent=1, sort_keys=True)
    for i, (B, P, Cin, Cout, gn) in enumerate(LAYERS):
        t = tab["conv_layer:%d:%d:%d:%s" % (Cin, Cout, B * P, WL)]
        hbm = 2.0 * t["fetch_size_kb"] + t["write_size_kb"]
        print("%5d -> %5d over %7d rows: fetch %10.0f KB x2 + write %10.0f KB = %7.3f GB = %.2f x algorithmic %.3f GB" % (
            Cin, Cout, B * P, t["fetch_size_kb"], t["write_size_kb"], hbm * 1024 / 1e9, hbm / t["algorithmic_kb"], t["algorithmic_kb"] * 1024 / 1e9))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        join(sys.argv[2], sys.argv[3], sys.argv[4])
