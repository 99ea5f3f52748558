#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
export ACL_SKIP_C5_FULL=1
timeout 1200 python -m pytest tests/test_sharded_gpu.py tests/test_callers_gpu.py tests/test_write_path_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r02_3_tests.log 2>&1; echo "tests rc=$?"
tail -25 $O/r02_3_tests.log
ACL_LOCAL_MAX=0 timeout 900 bash tools/ab.sh 2>&1 | tee $O/r02_3_ab.txt
timeout 300 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | tee $O/r02_3_batcher.txt
