#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_base.so timeout 600 bash tools/pmc_multi2.sh r02_pmc2_base 2>&1 | tee $O/r02_11_pmc2_base.txt
