#!/bin/bash
# round 4, call 17: the string entry point's repeated-name memo -- the string tests, then acl_check_bulk_v on the proxy's own batch shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py -m gpu -q --timeout 600 -k "string" 2>&1 | tail -3 | cut -c1-400
timeout 900 python tools/string_shapes.py 2>&1 | tail -1 | cut -c1-900
