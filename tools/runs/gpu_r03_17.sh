#!/bin/bash
# round 3, call 17: native sharded loop -- LookupResources, all-to-all, planned entry exchanges -- on logical shards
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 900 python -m pytest tests/test_sharded_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_17_tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/r03_17_tests.log
