#!/bin/bash
# round 3, call 22: k_check_queue with the commit moved behind the next pair's entry load
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACL_LOCAL_QUEUE=1 timeout -s KILL 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_22_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -3 $O/r03_22_tests.log
[ $rc -ne 0 ] && exit 0
run() {
  timeout -s KILL 200 python bench.py --workload ${W:-C4} --no-cpu --legs device --configs off --strings off --steps 30 2>$O/r03_22_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', round(d['value']/1e6,1), 'M/s kernel us', round(d['roofline']['kernel_avg_us'],1))"
}
ACL_LOCAL_QUEUE=0 run local
ACL_LOCAL_QUEUE=1 run queue
ACL_LOCAL_QUEUE=0 run local
ACL_LOCAL_QUEUE=1 run queue
W=C2 ACL_LOCAL_QUEUE=0 run localC2
W=C2 ACL_LOCAL_QUEUE=1 run queueC2
