#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_phases.so timeout 120 python tools/phases.py C4 2>&1 | tail -13 | tee $R/gpurun_out/r02_42_phases.txt
