#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/spicedb-kubeapi-proxy_amd/lib
run() { echo "== $*"; env "$@" bash tools/levels.sh r02_56_x 2>&1 | grep "last levels"; env "$@" bash tools/levels.sh r02_56_y --workload C2 2>&1 | grep "last levels"; }
run A=default
run ACLGPU_LIB=$L/libaclgpu_w3w7.so
run A=default
run ACLGPU_LIB=$L/libaclgpu_w3w7.so
