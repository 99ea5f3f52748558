#!/bin/bash
# usage (GPU box): tools/ab_variants.sh <workload> <rounds> <launches> [variant tags...] -- same-box A/B of lib/libaclgpu_<tag>.so ("main" = libaclgpu.so),
# the variants taken round-robin so that clock / thermal drift hits all of them alike (tools/ab_kernel.py prints one line per run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; ROUNDS=$2; K=$3; shift 3
for r in $(seq 1 $ROUNDS); do
  for T in "$@"; do
    L=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_$T.so
    [ "$T" = main ] && L=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu.so
    ACLGPU_LIB=$L timeout 600 python $R/tools/ab_kernel.py $W $K 2>&1 | tail -1
  done
done
