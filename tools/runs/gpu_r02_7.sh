#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
ACL_DEBUG_REBUILD=1 timeout 120 python -m pytest tests/test_write_path_gpu.py -m gpu -q --tb=line -p no:cacheprovider -s -k compaction > $O/r02_7_compaction.log 2>&1; echo "compaction test rc=$?"
grep -c "synchronous rebuild" $O/r02_7_compaction.log; grep "synchronous rebuild" $O/r02_7_compaction.log | head -8; tail -4 $O/r02_7_compaction.log
timeout 200 python -m pytest tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "cancellation or batcher" > $O/r02_7_tests.log 2>&1; echo "tests rc=$?"
tail -6 $O/r02_7_tests.log
