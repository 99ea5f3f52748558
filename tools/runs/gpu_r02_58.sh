#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
for SET in "sqa:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "sqi:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "grbm:GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=${SET%%:*}; C=${SET#*:}
  PMC_TIMEOUT=60 bash tools/pmc.sh r02_pmc_final/$N "$C" > /dev/null 2>&1
  DB=$(ls $O/prof/r02_pmc_final/$N/*.db 2>/dev/null | head -1)
  echo "== $N: $C"
  if [ -n "$DB" ]; then python tools/pmc_show.py $DB 3 k_check_local; else tail -3 $O/prof/r02_pmc_final/$N/run.log; fi
done 2>&1 | tee $O/r02_58_pmc_final.txt
