#!/bin/bash
# round 3, call 20: rocprofv3 evidence with the round's final kernels -- kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes for C4, C2, C3 and the beyond-L3 replica (C5R),
# issue / cache counter passes for C3, C2 and C4 (each --pmc set in a run of its own, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
rm -rf $O/prof/r03_*
bash tools/prof_c4.sh r03_c4 | cut -c1-200
bash tools/prof_c4.sh r03_c2 --workload C2 | cut -c1-200
bash tools/prof_c4.sh r03_c3 --workload C3 | cut -c1-200
bash tools/prof_c4.sh r03_c5r --workload C5 --replica | cut -c1-200
for W in C3 C2 C4; do
  echo "######## $W"; PMC_TIMEOUT=120 bash tools/pmc_multi.sh r03_pmc_$W --workload $W 2>&1 | tee $O/r03_20_pmc_$W.txt | tail -40
done
du -sh $O/prof
