#!/bin/bash
# kernel stats of the training step alone (cfg-3 shard): gpurun_out/<tag>_train_full_kernel_stats.txt
TAG=${1:-r04_run6}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pt && rocprofv3 --kernel-trace --stats -d /tmp/pt -o r -- python $REPO/bench_train.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pt.err
python $REPO/tools/rocprof_summary.py $(find /tmp/pt -name "*results.db" | head -1) $OUT/${TAG}_train_full_kernel_stats.txt
python $REPO/tools/train_host_probe.py > $OUT/${TAG}_train_host_probe.txt 2>&1
