#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_sharded_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r02_16_tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/r02_16_tests.log
timeout 600 bash tools/ab.sh 2>&1 | tee $O/r02_16_ab.txt
