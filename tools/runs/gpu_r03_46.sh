#!/bin/bash
# round 3, call 46: submitted tickets of host-mapped batches run as whole calls on the pool's workers -- tests that submit, then windows 1 ... 6
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 600 python -m pytest tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_engine_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_46_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r03_46_tests.log
run() {
  timeout -s KILL 300 python bench.py --workload $1 --no-cpu --configs off --strings off --steps 40 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 $2: value %.1f M/s | equal %s' % (d['value']/1e6, d['host_ids']['answers_equal_device_leg']))"
}
for w in 1 2 3 4 6; do run C4 "--pipeline submit --window $w"; done
for w in 2 4; do run C2 "--pipeline submit --window $w"; done
