#!/bin/bash
# round 4, call 11: the live-graph differential fuzz on the combine schema (exclusions, intersections, wildcards, a non-monotone userset subject):
# the regression tests, then longer campaigns
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --timeout 600 2>&1 | tail -4
for S in 31 32 33; do
  timeout 600 python tools/fuzz_gpu.py --schema combine --seed $S --steps 400 2>/dev/null | tail -1
done
timeout 600 python tools/fuzz_gpu.py --schema combine --seed 34 --steps 300 --compact-early 2>/dev/null | tail -1
timeout 600 python tools/fuzz_gpu.py --schema combine --seed 35 --steps 120 --burst 300 --universe 3 2>/dev/null | tail -1
