#!/bin/bash
# round 3, call 3: LookupResources batch-size sweep (kernel writes to host vs device rows + DMA vs level loop); rocprofv3 stats + FETCH/WRITE passes for C3 and C2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( echo "== k_rev_local, rows written to host memory by the kernel"; timeout 120 python tools/lookup_bench.py 1 4 16 64 256 1024
  echo "== k_rev_local, rows through a device buffer + one DMA copy (ACL_REV_ROWS=device)"; ACL_REV_ROWS=device timeout 120 python tools/lookup_bench.py 1 4 16 64 256 1024
  echo "== level loop (ACL_REV_LOCAL=0)"; ACL_REV_LOCAL=0 timeout 120 python tools/lookup_bench.py 1 64 1024 ) > $O/r03_3_lookup_sweep.txt 2>&1
cat $O/r03_3_lookup_sweep.txt | grep -v amdgpu.ids
bash tools/prof_c4.sh r03_c3 --workload C3 | cut -c1-400
bash tools/prof_c4.sh r03_c2 --workload C2 | cut -c1-400
ls $O/prof/r03_c3 $O/prof/r03_c3/stats | head -20
