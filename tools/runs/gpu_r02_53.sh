#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
L=$R/spicedb-kubeapi-proxy_amd/lib
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_sharded_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
run() { echo "== $*"; env "$@" bash tools/levels.sh r02_53_x 2>&1 | grep "last levels"; env "$@" bash tools/levels.sh r02_53_y --workload C2 2>&1 | grep "last levels"; }
run A=default
run ACLGPU_LIB=$L/libaclgpu_prev.so
run A=default
run ACLGPU_LIB=$L/libaclgpu_prev.so
