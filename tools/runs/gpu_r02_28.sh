#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_phases.so timeout 120 python tools/phases.py C4 2>&1 | tail -14 | tee $O/r02_28_phases.txt
ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_phases.so timeout 120 python tools/phases.py C2 2>&1 | tail -14 | tee -a $O/r02_28_phases.txt
