#!/bin/bash
# every concurrent call shape at once (tools/engine_stress.cpp), then the full GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 40 tools/bin/engine_stress 4 > $O/r02_64_engine_stress.txt 2>&1; echo "engine_stress rc=$?" | tee -a $O/r02_64_engine_stress.txt
tail -2 $O/r02_64_engine_stress.txt
timeout 118 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/r02_64_tests.log 2>&1; echo "tests rc=$?"
tail -2 $O/r02_64_tests.log
