#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 200 bash tools/pmc_sq.sh r02_pmc_walk 2>&1 | tee $O/r02_23_pmc_walk.txt
for U in 1 2; do
  echo "== C2 ACL_LOCAL_UPW=$U"; ACL_LOCAL_UPW=$U bash tools/levels.sh r02_c2_upw$U --workload C2 2>&1 | grep "last levels"
  echo "== C4 ACL_LOCAL_UPW=$U"; ACL_LOCAL_UPW=$U bash tools/levels.sh r02_c4_upw$U 2>&1 | grep "last levels"
done 2>&1 | tee $O/r02_23_upw.txt
