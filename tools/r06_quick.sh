#!/bin/bash
# quick GPU pass of round 6: index-kernel tests, headline with / without the guard, kernel stats + step timeline, per-layer conv traffic
TAG=${1:-r06b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fps or ball_query or accuracy_guard or calibrate or three_nn" > $OUT/tests.log 2>&1; echo "pytest rc=$?" >> $OUT/tests.log
cd /tmp
python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-subblock --no-sub-blocks > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?" >> $OUT/bench_quick.err
rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-subblock --no-sub-blocks --no-guard-subblock > $OUT/prof_bench.json 2> /tmp/pk.err
python $REPO/tools/rocprof_summary.py $(find /tmp/pk -name "*results.db" | head -1) $OUT/kernel_stats.txt
python $REPO/tools/rocprof_timeline.py $(find /tmp/pk -name "*results.db" | head -1) $OUT/step_timeline.txt
: > $OUT/conv_layers_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc && rocprofv3 --kernel-trace --pmc $C -d /tmp/pc -o r -- python $REPO/tools/conv_layers_pmc.py run > /dev/null 2> /tmp/pc.err
  rm -f /tmp/conv_$C.txt
  python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/pc -name "*results.db" | head -1) /tmp/conv_$C.txt --dispatches "conv1x1|chamfer_kernel|conv_gn"
  grep -E "^D |counter" /tmp/conv_$C.txt >> $OUT/conv_layers_pmc.txt
  tail -3 /tmp/pc.err >> $OUT/conv_layers_pmc.err
done
tail -4 $OUT/tests.log; tail -2 $OUT/bench_quick.err; wc -l $OUT/conv_layers_pmc.txt
