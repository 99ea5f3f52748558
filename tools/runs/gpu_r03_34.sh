#!/bin/bash
# round 3, call 34: C5-size replica on the LEVEL LOOP (one k_expand per level): FETCH / WRITE / L2 hit-miss per level -- where the beyond-L3 traffic comes from
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ACL_LOCAL_MAX=0 PMC_TIMEOUT=200
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=r03_34_$(echo $c | cut -d' ' -f1)
  bash tools/pmc.sh $tag "$c" --workload C5 --replica > /dev/null 2>&1
  echo "== $c"
  python tools/pmc_show.py $(ls gpurun_out/prof/$tag/*.db | head -1) 9 k_expand
done
