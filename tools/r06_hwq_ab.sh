#!/bin/bash
# A/B: the number of hardware queues the HIP runtime multiplexes the streams of a process onto (GPU_MAX_HW_QUEUES, default 4): reconstruct() has
# the main stream, the index chain, the global PointNet, a set-abstraction scale, the latent solve, the draw's copy and the guard in flight
OUT=gpurun_out/${1:-r06d}
mkdir -p $OUT
for rep in 1 2; do
for q in ${QS:-4 8 6 12}; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-subblock --no-sub-blocks 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
g = r['accuracy_guard']
print('GPU_MAX_HW_QUEUES=$q rep $rep: %.3f ms/step (guard on), %.3f guard off, cnf launch %.3f ms, stages %s' % (r['ms_per_step'], g['ms_per_step_guard_off'], r['roofline']['launch_ms'], {k: v for k, v in r['stage_ms_per_step'].items() if k.startswith('enc')}))
" | tee -a $OUT/hwq_ab.txt
done
done
