#!/bin/bash
# round 3, call 7: every declared class is live from the build on -- write-path tests, then the reference's dual-write sequence on the 10 M graph across a compaction
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 400 python -m pytest tests/test_write_path_gpu.py tests/test_engine_gpu.py tests/test_callers_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_7_tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/r03_7_tests.log
timeout 600 python tools/dual_write_latency.py > $O/r03_7_dual_write.json 2> $O/r03_7_dual_write.err; echo "dual write rc=$?"
tail -3 $O/r03_7_dual_write.err | grep -v amdgpu.ids; cat $O/r03_7_dual_write.json
