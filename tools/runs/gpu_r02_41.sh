#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/spicedb-kubeapi-proxy_amd/lib
run() { echo "== $*"; env "$@" bash tools/levels.sh r02_41_x 2>&1 | grep "last levels"; }
for i in 1 2; do
run A=default_e6
run ACLGPU_LIB=$L/libaclgpu_e4.so
run ACLGPU_LIB=$L/libaclgpu_e2.so
done
