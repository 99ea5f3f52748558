#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ACL_SKIP_C5_FULL=1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -2
