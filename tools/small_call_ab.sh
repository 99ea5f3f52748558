#!/bin/bash
# usage (GPU box): tools/small_call_ab.sh <rounds> <tags...> -- tools/small_call_probe.py per library variant, round-robin
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
for r in $(seq 1 $N); do for T in "$@"; do L=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_$T.so; [ "$T" = main ] && L=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu.so
  echo "== $T"; ACLGPU_LIB=$L python $R/tools/small_call_probe.py 2>&1 | grep "n="; done; done
