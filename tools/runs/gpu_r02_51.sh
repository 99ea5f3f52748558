#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "overflow or branching" 2>&1 | tail -6
