mkdir -p gpurun_out/r04
python tools/step_kernels.py 2>/dev/null | grep -E "sa_mlp_max:9|sa_mlp_max:99"
python -m pytest tests/test_hip_parity.py tests/test_error_budget.py -m gpu -q 2>&1 | tail -3
cp gpurun_out/parity_report.json gpurun_out/r04/f64ref_parity_full.json
python bench.py --clouds random --batch 64 --seq-len 20 --num-pts 4096 --steps 2 --warmup 1 --no-f32-subblock > gpurun_out/r04/cfg5_f64ref.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/cfg5_f64ref.json').read().strip().splitlines()[-1]); p=d["cpu_baseline"]["parity"]
print("cfg5", d["value"], d["parity_ok"], p["x"], p["tnocs"])
PY
