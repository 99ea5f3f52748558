#!/bin/bash
# round 3, call 16: acl_lookup_one_submit / acl_lookup_completions on the GPU (callers tests), tools/string_path.py on the r02 graph (250 k named objects)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 500 python -m pytest tests/test_callers_gpu.py tests/test_abi.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/string_path.py 2>/dev/null | tee $O/r03_16_string_path.json
