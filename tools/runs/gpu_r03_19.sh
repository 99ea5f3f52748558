#!/bin/bash
# round 3, call 19: the whole GPU suite after the sharded / string / list-filter / parity additions
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_19_tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/r03_19_tests.log
