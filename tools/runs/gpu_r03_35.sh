#!/bin/bash
# round 3, call 35: per-request membership filters in k_check_local -- parity (full-scale tests), then A/B on C4 and on the C5-size replica
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 600 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_35_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 $O/r03_35_tests.log
[ $rc -ne 0 ] && exit 0
run() {
  timeout -s KILL 400 python bench.py --workload $1 $2 --no-cpu --legs device --configs off --strings off --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 $2 filter=$ACL_LOCAL_FILTER value', round(d['value']/1e6,1), 'M/s kernel us', round(d['roofline']['kernel_avg_us'],1))"
}
for f in 0 1 0 1; do ACL_LOCAL_FILTER=$f run C4; done
for f in 0 1 0 1; do ACL_LOCAL_FILTER=$f run C5 --replica; done
for f in 0 1; do ACL_LOCAL_FILTER=$f run C2; done
