#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/r02_18_tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/r02_18_tests.log
ACL_LOCAL_MAX=300000 timeout 300 python -m pytest tests/test_fullscale_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "c4 or c2" > $O/r02_18_tests_local.log 2>&1; echo "tests(local for all sizes) rc=$?"
tail -3 $O/r02_18_tests_local.log
for CFG in "8192 2" "300000 1" "300000 2" "300000 3" "300000 4" "300000 8"; do
  set -- $CFG
  echo "== ACL_LOCAL_MAX=$1 ACL_LOCAL_UPW=$2"
  ACL_LOCAL_MAX=$1 ACL_LOCAL_UPW=$2 bash tools/levels.sh r02_local_$1_$2 2>&1 | grep "last levels"
  ACL_LOCAL_MAX=$1 ACL_LOCAL_UPW=$2 timeout 120 python bench.py --no-cpu --steps 20 --configs off --legs device 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('device-resident M/s', round(d['device_resident']['decisions_per_s']/1e6,1), 'kernel_ms', round(d['device_resident']['kernel_ms_per_batch'],4), d['device_resident'].get('expand_launches_per_batch'))"
done 2>&1 | tee $O/r02_18_local.txt
