mkdir -p gpurun_out/r04
for m in 1024 512 128; do
  CASPR_DEBUG=1 CASPR_X6W_MIN_CIN=$m python bench.py --no-cpu-baseline --no-sub-blocks --no-f32-subblock --steps 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('min_cin $m:', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernels'][1]['all_layers'], d['roofline']['kernels'][1]['frac'])"
done
