#!/bin/bash
# Round profile of the default path on the GPU box: kernel stats of bench.py (cfg-2), HBM-side traffic counters (FETCH_SIZE /
# WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes), the cfg-5 shape, and the training step.
# usage (from the repo root on the GPU box): bash tools/profile_round.sh r02     -> gpurun_out/<tag>_*
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $REPO/bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $OUT/${TAG}_bench.err
rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-f32-subblock --no-sub-blocks --no-guard-subblock > $OUT/${TAG}_prof_bench.json 2> /tmp/pk.err
python $REPO/tools/rocprof_summary.py $(find /tmp/pk -name "*results.db" | head -1) $OUT/${TAG}_kernel_stats.txt
python $REPO/tools/rocprof_timeline.py $(find /tmp/pk -name "*results.db" | head -1) $OUT/${TAG}_step_timeline.txt
: > $OUT/${TAG}_traffic_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pp && rocprofv3 --kernel-trace --pmc $C -d /tmp/pp -o r -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-subblock --no-sub-blocks --no-guard-subblock > /dev/null 2> /tmp/pp.err
  python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/pp -name "*results.db" | head -1) /tmp/pmc_$C.txt
  grep -E "counter|cnf_rk4|conv1x1_bf16x6|conv1x1_x6w|sa_mlp|sa_small|sa_repair|gn_partial" /tmp/pmc_$C.txt >> $OUT/${TAG}_traffic_pmc.txt
  rm -f /tmp/pmc_$C.txt
done
rm -rf /tmp/pm && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pm -o r -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-subblock --no-sub-blocks --no-guard-subblock > /dev/null 2> /tmp/pm.err
python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/pm -name "*results.db" | head -1) /tmp/pmc_sq.txt
grep -E "counter|cnf_rk4|conv1x1_bf16x6|conv1x1_x6w" /tmp/pmc_sq.txt > $OUT/${TAG}_sq_pmc.txt
# the traffic table bench.py reads (profiles/kernel_traffic.json): cfg-2
python $REPO/tools/make_traffic_table.py $OUT/${TAG}_traffic_pmc.txt 16x10x2048 8 ${TAG}_traffic_pmc.txt
# per-LAYER traffic of the pointwise convs (each layer of cfg-2 on its own between marker launches: tools/conv_layers_pmc.py)
: > $OUT/${TAG}_conv_layers_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc && rocprofv3 --kernel-trace --pmc $C -d /tmp/pc -o r -- python $REPO/tools/conv_layers_pmc.py run > /dev/null 2> /tmp/pc.err
  rm -f /tmp/conv_$C.txt
  python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/pc -name "*results.db" | head -1) /tmp/conv_$C.txt --dispatches "conv1x1|chamfer_kernel|conv_gn"
  grep -E "^D |counter" /tmp/conv_$C.txt >> $OUT/${TAG}_conv_layers_pmc.txt
done
python $REPO/tools/conv_layers_pmc.py join /tmp/conv_FETCH_SIZE.txt /tmp/conv_WRITE_SIZE.txt ${TAG}_conv_layers_pmc.txt > $OUT/${TAG}_conv_layers_traffic.txt 2>&1
# cfg-5 shape (BASELINE.json configs[4], one GPU's share): 64 sequences x 20 x 4096, random clouds
python $REPO/bench.py --clouds random --batch 64 --seq-len 20 --num-pts 4096 --steps 3 --warmup 1 --no-f32-subblock > $OUT/${TAG}_cfg5_bench.json 2> $OUT/${TAG}_cfg5_bench.err; echo "cfg5 rc=$?" >> $OUT/${TAG}_cfg5_bench.err
rm -rf /tmp/p5 && rocprofv3 --kernel-trace --stats -d /tmp/p5 -o r -- python $REPO/bench.py --clouds random --batch 64 --seq-len 20 --num-pts 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-f32-subblock --no-guard-subblock > /dev/null 2> /tmp/p5.err
python $REPO/tools/rocprof_summary.py $(find /tmp/p5 -name "*results.db" | head -1) $OUT/${TAG}_cfg5_kernel_stats.txt
: > $OUT/${TAG}_cfg5_traffic_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pq && rocprofv3 --kernel-trace --pmc $C -d /tmp/pq -o r -- python $REPO/bench.py --clouds random --batch 64 --seq-len 20 --num-pts 4096 --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subblock --no-guard-subblock > /dev/null 2> /tmp/pq.err
  python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/pq -name "*results.db" | head -1) /tmp/pmq_$C.txt
  grep -E "counter|cnf_rk4|conv1x1_bf16x6|conv1x1_x6w" /tmp/pmq_$C.txt >> $OUT/${TAG}_cfg5_traffic_pmc.txt
  rm -f /tmp/pmq_$C.txt
done
python $REPO/tools/make_traffic_table.py $OUT/${TAG}_cfg5_traffic_pmc.txt 64x20x4096 8 ${TAG}_cfg5_traffic_pmc.txt
# the stress regime's step count (S = 64: bench.py's stress_dynamics sub-block prices the same kernel at that count)
: > $OUT/${TAG}_s64_traffic_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ps && rocprofv3 --kernel-trace --pmc $C -d /tmp/ps -o r -- python $REPO/bench.py --cnf-steps 64 --steps 1 --warmup 1 --no-cpu-baseline --no-f32-subblock --no-sub-blocks --no-guard-subblock > /dev/null 2> /tmp/ps.err
  python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/ps -name "*results.db" | head -1) /tmp/pms_$C.txt
  grep -E "counter|cnf_rk4" /tmp/pms_$C.txt >> $OUT/${TAG}_s64_traffic_pmc.txt
  rm -f /tmp/pms_$C.txt
done
python $REPO/tools/make_traffic_table.py $OUT/${TAG}_s64_traffic_pmc.txt 16x10x2048 64 ${TAG}_s64_traffic_pmc.txt
# training step (cfg-3 shard)
python $REPO/bench_train.py --steps 3 --warmup 1 > $OUT/${TAG}_bench_train_full.json 2> $OUT/${TAG}_bench_train.err
python $REPO/bench_train.py --steps 3 --warmup 1 --mode pretrain --no-cpu-baseline > $OUT/${TAG}_bench_train_pretrain.json 2>> $OUT/${TAG}_bench_train.err
rm -rf /tmp/pt && rocprofv3 --kernel-trace --stats -d /tmp/pt -o r -- python $REPO/bench_train.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pt.err
python $REPO/tools/rocprof_summary.py $(find /tmp/pt -name "*results.db" | head -1) $OUT/${TAG}_train_full_kernel_stats.txt
# HBM-side traffic of the training step's matrix / activation kernels (a counter pass over ALL ~4,000 launches of a step does not finish)
: > $OUT/${TAG}_train_traffic_pmc.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pz && timeout 900 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "conv1x1_bf16x6_kernel|conv1x1_wgrad_bf16x6_kernel|conv1x1_x6w_kernel|cnf_act_bwd_kernel|cnf_in_.*_rows_kernel|conv1x1_stream_kernel" -d /tmp/pz -o r -- python $REPO/bench_train.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pz.err
  python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/pz -name "*results.db" | head -1) /tmp/pmz_$C.txt
  grep -E "counter|conv1x1|cnf_" /tmp/pmz_$C.txt >> $OUT/${TAG}_train_traffic_pmc.txt
  rm -f /tmp/pmz_$C.txt
done
CASPR_PMC_TRAIN=1 python $REPO/tools/make_traffic_table.py $OUT/${TAG}_train_traffic_pmc.txt 8x10x1024 8 ${TAG}_train_traffic_pmc.txt
# the CNF kernel alone (kernel stats + SQ counters) and the two micro-benchmarks behind DESIGN.md's power / filler discussion
(cd $REPO && bash tools/profile_cnf.sh ${TAG} > /dev/null 2>&1)
[ -x $REPO/tools/micro/mfma_power ] && $REPO/tools/micro/mfma_power > $OUT/${TAG}_mfma_power.txt 2>&1
[ -x $REPO/tools/micro/mfma_fillers ] && $REPO/tools/micro/mfma_fillers > $OUT/${TAG}_mfma_fillers.txt 2>&1
[ -x $REPO/tools/micro/mfma_pipe ] && $REPO/tools/micro/mfma_pipe 800 > $OUT/${TAG}_mfma_pipe.txt 2>&1
[ -x $REPO/tools/micro/mfma_dma ] && $REPO/tools/micro/mfma_dma > $OUT/${TAG}_mfma_dma.txt 2>&1
python $REPO/tools/host_timeline.py > $OUT/${TAG}_host_timeline.txt 2>&1
python $REPO/tools/train_host_probe.py > $OUT/${TAG}_train_host_probe.txt 2>&1
python $REPO/tools/sa_mlp_phase_trace.py > $OUT/${TAG}_sa_mlp_phase_trace.txt 2>&1
python $REPO/tools/conv_bench.py > $OUT/${TAG}_conv_bench.txt 2>&1
# per-chunk phase cycles of the bf16x6 conv (debug flavour of the library, if it was built) and weight-gradient timings in both modes
if [ -f $REPO/caspr_amd/csrc/libcaspr_hip_debug.so ]; then python $REPO/tools/conv_x6_trace.py > $OUT/${TAG}_conv_x6_trace.txt 2>&1; fi
python $REPO/tools/wgrad_bench.py > $OUT/${TAG}_wgrad_bench.txt 2>&1
ls -la $OUT | grep ${TAG}_ | head -30
