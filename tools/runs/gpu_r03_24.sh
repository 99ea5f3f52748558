#!/bin/bash
# round 3, call 24: units per resident block (in-flight requests per XCD vs the L2) on C4 and on the C5-size replica
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
run() {
  timeout -s KILL 400 python bench.py --workload $1 $2 --no-cpu --legs device --configs off --strings off --steps 20 2>$O/r03_24_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 $2 upw=$ACL_LOCAL_UPW wide_min=$ACL_LOCAL_WIDE_MIN value', round(d['value']/1e6,1), 'M/s kernel us', round(d['roofline']['kernel_avg_us'],1))"
}
for u in 1 2 4; do ACL_LOCAL_UPW=$u run C4; done
for u in 1 2 4; do ACL_LOCAL_UPW=$u run C5 --replica; done
