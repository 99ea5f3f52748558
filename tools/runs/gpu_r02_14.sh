#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 100 python -m pytest tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
timeout 200 bash tools/pmc_sq.sh r02_pmc_v12 2>&1 | tee $O/r02_14_pmc_v12.txt
