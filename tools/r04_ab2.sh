mkdir -p gpurun_out/r04
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-sub-blocks --no-f32-subblock --steps 10 2>gpurun_out/r04/ab2_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run:', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
tail -3 gpurun_out/r04/ab2_err.txt
