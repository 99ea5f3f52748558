#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_w3.so timeout 200 bash tools/pmc_sq.sh r02_pmc_walk2 2>&1 | tee $O/r02_26_pmc_walk2.txt
