#!/usr/bin/env python
"""profiles/kernel_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh: HBM-side KB per launch of the
kernels bench.py prices, keyed by kernel + workload shape (bench.py reads it: the counters cannot be read from inside the
benchmark process).   usage: tools/make_traffic_table.py <traffic_pmc.txt> <BxTxN> <cnf_steps> <committed name of that file>"""
import json, os, re, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src, wl, steps, committed = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rows = {}
for ln in open(src):
    m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([0-9.e+]+)\s+([0-9.e+]+)\s+([0-9.e+]+)\s*$", ln.rstrip("\n"))
    if m:
        rows.setdefault(m.group(1).strip(), {})[m.group(2)] = {"n": int(m.group(3)), "avg": float(m.group(4)), "min": float(m.group(5)), "max": float(m.group(6))}
path = os.path.join(ROOT, "profiles", "kernel_traffic.json")
tab = json.load(open(path)) if os.path.exists(path) else {}
note = "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> x2; separate rocprofv3 --pmc passes of `python bench.py --steps 2 --warmup 1` (tools/profile_round.sh)"
B, T, N = (int(v) for v in wl.split("x"))
for k, v in rows.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    if k.startswith("cnf_rk4_x6w_kernel") or k.startswith("void cnf_rk4_kernel<false>"):
        name = "cnf_rk4_x6w_kernel" if "x6w" in k else "cnf_rk4_kernel"
        tab["%s:%s:s%d" % (name, wl, steps)] = {"fetch_size_kb_per_launch": v["FETCH_SIZE"]["avg"], "write_size_kb_per_launch": v["WRITE_SIZE"]["avg"],
                                                "fetch_correction": 2.0, "source": "profiles/%s" % committed, "note": note}
# the largest pointwise conv (the 1600 -> 1600 head layer): on conv1x1_x6w_kernel when that kernel ran (its only launch per step),
# else the largest launch of conv1x1_bf16x6_kernel
xw = [(v["FETCH_SIZE"]["avg"], v["WRITE_SIZE"]["avg"]) for k, v in rows.items() if "conv1x1_x6w_kernel" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
big = xw or [(v["FETCH_SIZE"]["max"], v["WRITE_SIZE"]["max"]) for k, v in rows.items() if "conv1x1_bf16x6_kernel" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
if big:
    f, w = max(big)
    tab["conv_largest:1600x1600:%s" % wl] = {"fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w, "fetch_correction": 2.0,
                                                    "source": "profiles/%s" % committed,
                                                    "kernel": "conv1x1_x6w_kernel (first 1536 channels; the 64-channel remainder's launch on conv1x1_bf16x6_kernel is not included)" if xw else "conv1x1_bf16x6_kernel",
                                                    "note": note + "; the 1600 -> 1600 head layer over %d rows (input %.2f GB + output %.2f GB algorithmic)"
                                                            % (B * T * N, B * T * N * 1600 * 4 / 1e9, B * T * N * 1600 * 4 / 1e9)}
# the fused set-abstraction launches (sa_small_kernel / sa_mlp_kernel, every instantiation): HBM-side KB per STEP = sum over the kernels
# of (average per launch x launches per step); the passes above ran `--steps 2 --warmup 1` = 3 steps + 2 detail steps = 5 steps
sa = [(k, v) for k, v in rows.items() if ("sa_small_kernel" in k or "sa_mlp_kernel" in k or "sa_repair_f64_kernel" in k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
if sa:
    steps_run = int(os.environ.get("CASPR_PMC_STEPS", "5"))
    f = sum(v["FETCH_SIZE"]["avg"] * v["FETCH_SIZE"]["n"] for _, v in sa) / steps_run
    w = sum(v["WRITE_SIZE"]["avg"] * v["WRITE_SIZE"]["n"] for _, v in sa) / steps_run
    tab["sa_all:%s" % wl] = {"fetch_size_kb_per_step": f, "write_size_kb_per_step": w, "fetch_correction": 2.0, "launches_counted": sum(v["FETCH_SIZE"]["n"] for _, v in sa),
                             "steps_in_the_pass": steps_run, "source": "profiles/%s" % committed, "note": note + "; all fused set-abstraction launches of one step together"}
# training step (tools/profile_round.sh: counters collected for the matrix / activation kernels of the CNF block and the convs only,
# --kernel-include-regex; `bench_train.py --steps 1 --warmup 1` = 3 steps with the detail pass): HBM-side KB per STEP over those kernels
if os.environ.get("CASPR_PMC_TRAIN"):
    steps_run = int(os.environ.get("CASPR_PMC_STEPS", "3"))
    ok = [(k, v) for k, v in rows.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    if ok:
        tab["train_matrix_kernels:%s" % wl] = {"fetch_size_kb_per_step": sum(v["FETCH_SIZE"]["avg"] * v["FETCH_SIZE"]["n"] for _, v in ok) / steps_run,
                                               "write_size_kb_per_step": sum(v["WRITE_SIZE"]["avg"] * v["WRITE_SIZE"]["n"] for _, v in ok) / steps_run,
                                               "fetch_correction": 2.0, "launches_counted": sum(v["FETCH_SIZE"]["n"] for _, v in ok), "steps_in_the_pass": steps_run,
                                               "kernels": sorted(k[:60] for k, _ in ok), "source": "profiles/%s" % committed,
                                               "note": note.replace("bench.py --steps 2 --warmup 1", "bench_train.py --steps 1 --warmup 1 --no-cpu-baseline") + "; only the kernels listed (rocprofv3 --kernel-include-regex): a counter pass over all ~4,000 launches of a step does not finish in the box's time"}
json.dump(tab, open(path, "w"), indent=1, sort_keys=True)
print("wrote %s: %s" % (path, sorted(tab)))
