#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
ACL_LOCAL_MAX=300000 timeout 300 python -m pytest tests/test_fullscale_gpu.py tests/test_engine_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/r02_19_tests_local.log 2>&1; echo "tests(local for all sizes) rc=$?"
tail -3 $O/r02_19_tests_local.log
for CFG in "8192 2 262144" "300000 2 262144" "300000 1 262144" "8192 2 65536" "300000 2 65536" "8192 2 16384" "300000 2 16384"; do
  set -- $CFG
  echo "== ACL_LOCAL_MAX=$1 ACL_LOCAL_UPW=$2 batch $3"
  ACL_LOCAL_MAX=$1 ACL_LOCAL_UPW=$2 timeout 120 python bench.py --no-cpu --steps 20 --configs off --legs device --batch $3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('device-resident M/s', round(d['device_resident']['decisions_per_s']/1e6,1), 'ms/batch', round(d['device_resident']['ms_per_batch'],4), 'kernel_ms', round(d['device_resident']['kernel_ms_per_batch'],4), d['device_resident'].get('expand_launches_per_batch'))"
done 2>&1 | tee $O/r02_19_local.txt
