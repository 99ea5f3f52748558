#!/bin/bash
# round 3, call 5: k_rev_local with workgroup-scope atomics / fences: parity tests, then the batch-size sweep again
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 300 python -m pytest tests/test_lookup_local_gpu.py tests/test_fullscale_gpu.py::test_c3_full_all_power_users tests/test_engine_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_5_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r03_5_tests.log
( echo "== k_rev_local (workgroup scope, result rows in LDS), rows written to host memory by the kernel"; timeout 120 python tools/lookup_bench.py 1 4 16 64 256 1024
  echo "== ... rows through a device buffer + one DMA copy (ACL_REV_ROWS=device)"; ACL_REV_ROWS=device timeout 120 python tools/lookup_bench.py 1 64 256 1024 ) > $O/r03_5_lookup_sweep.txt 2>&1
grep -v amdgpu.ids $O/r03_5_lookup_sweep.txt
( echo "== ... result rows in HBM (ACL_REV_LDS_ROWS=0)"; ACL_REV_LDS_ROWS=0 timeout 120 python tools/lookup_bench.py 1 64 256 ) 2>&1 | grep -v amdgpu.ids | tee -a $O/r03_5_lookup_sweep.txt
for i in 1 2; do timeout 120 python bench.py --workload C3 --no-cpu --steps 40 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('C3 lookups/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['kernel_ms_per_step'],4), 'single p50 ms', round(d['p50_single_lookup_ms'],4), 'pageable', d['pageable_result_buffers'])"; done
