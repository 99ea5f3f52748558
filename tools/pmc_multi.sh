#!/bin/bash
# usage (GPU box): tools/pmc_multi.sh <tag> -- runs several single-purpose --pmc passes of bench.py and prints per-level tables
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for SET in "sq1:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "grbm:GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=${SET%%:*}; C=${SET#*:}
  bash $R/tools/pmc.sh $TAG/$N "$C" "$@" > /dev/null 2>&1
  DB=$(ls $R/gpurun_out/prof/$TAG/$N/*.db 2>/dev/null | head -1)
  echo "== $N: $C"
  if [ -n "$DB" ]; then python $R/tools/pmc_show.py $DB 7; else tail -5 $R/gpurun_out/prof/$TAG/$N/run.log; fi
done
