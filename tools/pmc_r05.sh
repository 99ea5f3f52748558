#!/bin/bash
# usage (GPU box): tools/pmc_r05.sh <C4|C5R> -- the counter passes VERDICT r4 next #1 asks for, on the FINAL kernel: one rocprofv3 --pmc pass per
# set (own run, --kernel-trace only; the vector-L1 set holds two counters: the five-counter TCP set aborts rocprofv3 on this pool), summarised by tools/pmc_r05_summary.py into gpurun_out/r05_pmc_<cfg>.md
CFG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=""
[ "$CFG" = C5R ] && ARGS="--workload C5 --replica"
TAG=r05_pmc_$CFG
i=0
for SET in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PMC_TIMEOUT=150 bash $R/tools/pmc.sh $TAG/s$i "$SET" $ARGS > /dev/null 2>&1
done
python $R/tools/pmc_r05_summary.py $R/gpurun_out/prof/$TAG $CFG > $R/gpurun_out/r05_pmc_$(echo $CFG | tr A-Z a-z).md
cat $R/gpurun_out/r05_pmc_$(echo $CFG | tr A-Z a-z).md
