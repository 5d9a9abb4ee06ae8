# HBM-side traffic of the 1600 -> 1600 head layer alone on the persistent 512-channel kernel: with the default dispatch the kernel also
# runs other layers and the head layer in two pieces, which the per-kernel PMC summary cannot tell apart -> this pass restricts the
# kernel to that layer (CASPR_X6W_MIN_CIN=1024) in one piece (CASPR_EARLY_LATENT=0)
export TMPDIR=/tmp CASPR_DEBUG=1 CASPR_X6W_MIN_CIN=1024 CASPR_EARLY_LATENT=0
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
: > $OUT/r04_run3_conv_traffic_pmc.txt
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pp && rocprofv3 --kernel-trace --pmc $C -d /tmp/pp -o r -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-subblock --no-sub-blocks > /dev/null 2> /tmp/pp.err
  python $REPO/tools/rocprof_pmc_summary.py $(find /tmp/pp -name "*results.db" | head -1) /tmp/pmc_$C.txt
  grep -E "counter|conv1x1_x6w" /tmp/pmc_$C.txt >> $OUT/r04_run3_conv_traffic_pmc.txt
  rm -f /tmp/pmc_$C.txt
done
cat $OUT/r04_run3_conv_traffic_pmc.txt
python $REPO/tools/draw_contention.py > $OUT/r04_draw_contention.txt 2>&1; cat $OUT/r04_draw_contention.txt
