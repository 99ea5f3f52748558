#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r02_27_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r02_27_tests.log
run() { echo "== $*"; env "$@" bash tools/levels.sh r02_27_x 2>&1 | grep "last levels"; env "$@" bash tools/levels.sh r02_27_y --workload C2 2>&1 | grep "last levels"; }
run ACL_LOCAL_UPW=1
run ACL_LOCAL_UPW=2
run ACL_LOCAL_UPW=3
run ACL_LOCAL_UPW=4
