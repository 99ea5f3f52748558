#!/bin/bash
# usage (GPU box): tools/ab.sh -- bench (no CPU leg) + per-level k_expand durations for every lib/libaclgpu*.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for L in $R/spicedb-kubeapi-proxy_amd/lib/libaclgpu*.so; do
  echo "== $(basename $L)"
  ACLGPU_LIB=$L bash $R/tools/levels.sh ab_$(basename $L .so) 2>&1 | grep "last levels"
  ACLGPU_LIB=$L python $R/bench.py --no-cpu --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('M/s', round(d['value']/1e6,1), 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['kernel_ms_per_batch'],4))"
done
