#!/bin/bash
# round 3, call 12: where does the 16-wave instantiation of k_check_local start to pay?  (C4, device-resident, all sizes) + parity of both instantiations at full size
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( echo "== 4 waves per block at every size (ACL_LOCAL_WIDE_MIN=2000000000)"; ACL_LOCAL_WIDE_MIN=2000000000 timeout 200 python tools/batch_sweep.py
  echo "== 16 waves per block at every size (ACL_LOCAL_WIDE_MIN=0)"; ACL_LOCAL_WIDE_MIN=0 timeout 200 python tools/batch_sweep.py ) 2>&1 | grep -v amdgpu.ids | tee $O/r03_12_wide_threshold.txt
ACL_LOCAL_WIDE_MIN=0 timeout 300 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py::test_c4_full_every_entry_point tests/test_fullscale_gpu.py::test_c2_full -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
