#!/bin/bash
# round 3, call 5: k_rev_local with workgroup-scope atomics / fences: parity tests, then the batch-size sweep again
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 300 python -m pytest tests/test_lookup_local_gpu.py tests/test_fullscale_gpu.py::test_c3_full_all_power_users tests/test_engine_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_4_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r03_4_tests.log
( echo "== k_rev_local (workgroup scope, result rows in LDS), rows written to host memory by the kernel"; timeout 120 python tools/lookup_bench.py 1 4 16 64 256 1024
  echo "== ... rows through a device buffer + one DMA copy (ACL_REV_ROWS=device)"; ACL_REV_ROWS=device timeout 120 python tools/lookup_bench.py 1 64 256 1024 ) > $O/r03_4_lookup_sweep.txt 2>&1
grep -v amdgpu.ids $O/r03_4_lookup_sweep.txt
