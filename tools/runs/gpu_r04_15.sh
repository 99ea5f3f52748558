#!/bin/bash
# round 4, call 15: intersection arrows `.all()` -- the combine tests (hypothesis graphs against both oracles, nested / subtracted / intersected
# .all()), then the live-graph fuzz on the combine schema, which now has two permissions with .all()
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_combine_gpu.py tests/test_fuzz_gpu.py tests/test_engine_gpu.py -m gpu -q --timeout 600 2>&1 | tail -8 | cut -c1-600
for S in 41 42; do
  timeout 600 python tools/fuzz_gpu.py --schema combine --seed $S --steps 400 2>&1 | tail -1 | cut -c1-600
done
timeout 600 python tools/fuzz_gpu.py --schema combine --seed 43 --steps 300 --compact-early 2>&1 | tail -1 | cut -c1-600
