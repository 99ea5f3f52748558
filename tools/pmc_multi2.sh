#!/bin/bash
# (no TA_* / TD_* set: that pass hung rocprofv3 on this pool until its timeout)
# usage (GPU box): tools/pmc_multi2.sh <tag> -- second counter campaign: who holds the issue port, TA/TD busy, TCP stalls + TLB, LDS conflicts
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for SET in "sqa:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "sqi:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "sql:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY" \
           "tcp2:TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" \
           "tlb:TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_PENDING_STALL_CYCLES_sum"; do
  N=${SET%%:*}; C=${SET#*:}
  bash $R/tools/pmc.sh $TAG/$N "$C" "$@" > /dev/null 2>&1
  DB=$(ls $R/gpurun_out/prof/$TAG/$N/*.db 2>/dev/null | head -1)
  echo "== $N: $C"
  if [ -n "$DB" ]; then python $R/tools/pmc_show.py $DB 7; else tail -5 $R/gpurun_out/prof/$TAG/$N/run.log; fi
done
