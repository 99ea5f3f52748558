#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
for CFG in "C2 0 1" "C2 1048576 1" "C2 1048576 2" "C4 1048576 1" "C4 1048576 2" "C3 1048576 1"; do
  set -- $CFG
  echo "== $1 ACL_LOCAL_MAX=$2 ACL_LOCAL_UPW=$3"
  ACL_LOCAL_MAX=$2 ACL_LOCAL_UPW=$3 timeout 120 python bench.py --workload $1 --no-cpu --steps 30 --configs off --legs device 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('device_resident',{}); print('value', round(d['value']/1e6,3), d['unit'], '| kernel_ms', r.get('kernel_ms_per_batch'), r.get('dominant_kernel'), r.get('launches_per_batch'))"
done 2>&1 | tee $O/r02_21_upw.txt
