#!/bin/bash
# rocprofv3 passes on the CNF sampling kernel alone (tools/cnf_only.py): kernel stats + SQ / GRBM counters.
# usage (repo root, GPU box): bash tools/profile_cnf.sh <tag>   -> gpurun_out/<tag>_cnf_*
TAG=${1:-r03}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
python $REPO/tools/cnf_only.py 5 > $OUT/${TAG}_cnf_time.txt 2>&1
rm -rf /tmp/ck && rocprofv3 --kernel-trace --stats -d /tmp/ck -o r -- python $REPO/tools/cnf_only.py 3 > /dev/null 2> /tmp/ck.err
python $REPO/tools/rocprof_summary.py $(find /tmp/ck -name "*results.db" | head -1) $OUT/${TAG}_cnf_kernel_stats.txt
: > $OUT/${TAG}_cnf_pmc.txt
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/cp$i && rocprofv3 --kernel-trace --pmc $SET -d /tmp/cp$i -o r -- python $REPO/tools/cnf_only.py 2 > /dev/null 2> /tmp/cp$i.err
  DB=$(find /tmp/cp$i -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $REPO/tools/rocprof_pmc_summary.py $DB /tmp/cpmc$i.txt && grep -E "counter|cnf_rk4" /tmp/cpmc$i.txt >> $OUT/${TAG}_cnf_pmc.txt; else echo "pass $i failed: $(tail -3 /tmp/cp$i.err)" >> $OUT/${TAG}_cnf_pmc.txt; fi
done
cat $OUT/${TAG}_cnf_time.txt; grep -E "cnf_rk4|kernel" $OUT/${TAG}_cnf_kernel_stats.txt | head -5; cat $OUT/${TAG}_cnf_pmc.txt
