#!/bin/bash
# usage (GPU box): tools/pmc_sq.sh <tag> -- the two SQ passes that say who holds the issue port (subset of pmc_multi2.sh)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for SET in "sqa:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "sqi:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  N=${SET%%:*}; C=${SET#*:}
  PMC_TIMEOUT=60 bash $R/tools/pmc.sh $TAG/$N "$C" "$@" > /dev/null 2>&1
  DB=$(ls $R/gpurun_out/prof/$TAG/$N/*.db 2>/dev/null | head -1)
  echo "== $N: $C"
  if [ -n "$DB" ]; then python $R/tools/pmc_show.py $DB 7; else tail -5 $R/gpurun_out/prof/$TAG/$N/run.log; fi
done
