#!/bin/bash
# round 3, call 36: does the membership filter cut traffic?  FETCH_SIZE and L1->L2 requests of k_check_local, filter off / on, C5-size replica and C4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PMC_TIMEOUT=200
for wl in "--workload C5 --replica" "--workload C4"; do
for f in 0 1; do
for c in FETCH_SIZE "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_MISS_sum"; do
  tag=r03_36_${f}_$(echo $c | cut -d' ' -f1)
  rm -rf gpurun_out/prof/$tag
  ACL_LOCAL_FILTER=$f bash tools/pmc.sh $tag "$c" $wl > /dev/null 2>&1
  echo "== $wl filter=$f $c"
  python tools/pmc_show.py $(ls gpurun_out/prof/$tag/*.db | head -1) 2 k_check_local
done; done; done
