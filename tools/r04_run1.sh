set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
( time python bench.py ) > gpurun_out/r04/run1_bench.json 2> gpurun_out/r04/run1_bench.err
tail -c 600 gpurun_out/r04/run1_bench.err
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04/prof1 -o run1 -- python $R/bench.py --no-cpu-baseline --no-sub-blocks --no-f32-subblock --steps 5 --warmup 2 > $R/gpurun_out/r04/run1_prof_bench.json 2> $R/gpurun_out/r04/run1_prof.err
cd $R
DB=$(find gpurun_out/r04/prof1 -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r04/run1_kernel_stats.txt
rm -rf gpurun_out/r04/prof1
head -12 gpurun_out/r04/run1_kernel_stats.txt
