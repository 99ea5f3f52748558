#!/bin/bash
# usage (GPU box): tools/perturb.sh -- per-level k_expand durations of the baseline and of every ACL_PERTURB variant found in lib/
R=${GRAFT_REPO_ROOT:-$(pwd)}
for L in $R/spicedb-kubeapi-proxy_amd/lib/libaclgpu.so $R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_p*.so; do
  echo "== $(basename $L)"
  ACLGPU_LIB=$L bash $R/tools/levels.sh pert_$(basename $L .so) 2>&1 | grep "last levels"
done
