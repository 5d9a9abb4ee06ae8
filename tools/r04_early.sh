mkdir -p gpurun_out/r04
for e in 0 single team 0 team; do
  CASPR_DEBUG=1 CASPR_EARLY_LATENT=$e python bench.py --no-cpu-baseline --no-sub-blocks --no-f32-subblock --steps 8 2>gpurun_out/r04/early_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('early_latent $e:', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
