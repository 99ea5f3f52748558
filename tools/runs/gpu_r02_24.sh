#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
L=$R/spicedb-kubeapi-proxy_amd/lib
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_sharded_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r02_24_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r02_24_tests.log
run() { # name env...
  echo "== $*"
  env "$@" bash tools/levels.sh r02_24_x 2>&1 | grep "last levels"
  env "$@" bash tools/levels.sh r02_24_y --workload C2 2>&1 | grep "last levels"
}
run A=default
run ACLGPU_LIB=$L/libaclgpu_w3.so
run ACLGPU_LIB=$L/libaclgpu_w8.so ACL_PROG_LDS=0
run ACL_PROG_LDS=0
run ACL_LOCAL_MAX=0
run ACL_LOCAL_MAX=0 ACLGPU_LIB=$L/libaclgpu_w8.so ACL_PROG_LDS=0
