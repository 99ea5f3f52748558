#!/bin/bash
# usage (GPU box): tools/ab.sh -- bench (no CPU leg) + per-level k_expand durations for every lib/libaclgpu*.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for L in $R/spicedb-kubeapi-proxy_amd/lib/libaclgpu*.so; do
  echo "== $(basename $L)"
  ACLGPU_LIB=$L bash $R/tools/levels.sh ab_$(basename $L .so) 2>&1 | grep "last levels"
  ACLGPU_LIB=$L python $R/bench.py --no-cpu --steps 30 --configs off 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined host M/s', round(d['value']/1e6,1), '| device-resident M/s', round(d['device_resident']['decisions_per_s']/1e6,1), 'kernel_ms', round(d['device_resident']['kernel_ms_per_batch'],4), '| p50 host batch ms', round(d['latency']['p50_batch_ms'],4))"
done
