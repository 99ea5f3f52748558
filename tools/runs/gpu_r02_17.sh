#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
for LM in 8192 300000; do
  echo "== ACL_LOCAL_MAX=$LM"
  ACL_LOCAL_MAX=$LM bash tools/levels.sh r02_local_$LM 2>&1 | grep "last levels"
  ACL_LOCAL_MAX=$LM timeout 120 python bench.py --no-cpu --steps 20 --configs off --legs device 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('device-resident M/s', round(d['device_resident']['decisions_per_s']/1e6,1), 'kernel_ms', round(d['device_resident']['kernel_ms_per_batch'],4), d['device_resident'].get('expand_launches_per_batch'))"
done 2>&1 | tee $O/r02_17_local.txt
