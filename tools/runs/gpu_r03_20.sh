#!/bin/bash
# round 3, call 20: k_check_queue (no level barriers) -- parity first, then the same-box A/B against k_check_local on C4 and C2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACL_LOCAL_QUEUE=1 timeout -s KILL 420 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_20_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -6 $O/r03_20_tests.log
[ $rc -ne 0 ] && exit 0
for q in 0 1 0 1; do
  for w in C4 C2; do
    ACL_LOCAL_QUEUE=$q timeout -s KILL 200 python bench.py --workload $w --no-cpu --legs device --configs off --strings off --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('queue=$q $w value', round(d['value']/1e6,1), 'M/s kernel us', round(d['roofline']['kernel_avg_us'],1), d['roofline']['kernel'], 'parity', d.get('parity',{}).get('mismatches'))"
  done
done
