#!/bin/bash
# round 3, call 21: k_check_queue variants (geometry print, commit list size, sleep, inline) against k_check_local on C4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACL_LOCAL_QUEUE=1 timeout -s KILL 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_21_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -3 $O/r03_21_tests.log
[ $rc -ne 0 ] && exit 0
run() {
  timeout -s KILL 200 python bench.py --workload C4 --no-cpu --legs device --configs off --strings off --steps 30 2>$O/r03_21_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', round(d['value']/1e6,1), 'M/s kernel us', round(d['roofline']['kernel_avg_us'],1))"
  grep "single-launch walk" $O/r03_21_err.txt | head -2
}
ACL_DEBUG_GEOM=1 ACL_LOCAL_QUEUE=0 run local
ACL_DEBUG_GEOM=1 ACL_LOCAL_QUEUE=1 run queue
for v in qinl qsl8 qp31; do ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_$v.so ACL_LOCAL_QUEUE=1 run $v; done
ACL_LOCAL_WIDE_MIN=100000000 ACL_DEBUG_GEOM=1 ACL_LOCAL_QUEUE=1 run queue_narrow
ACL_LOCAL_WIDE_MIN=100000000 ACL_LOCAL_QUEUE=0 run local_narrow
