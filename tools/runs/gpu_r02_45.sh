#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/spicedb-kubeapi-proxy_amd/lib
echo "== new"; timeout 200 python tools/string_path.py 2>&1 | tail -1
echo "== prev"; ACLGPU_LIB=$L/libaclgpu_prev.so timeout 200 python tools/string_path.py 2>&1 | tail -1
