#!/bin/bash
# round 3, call 39: differential fuzz campaigns (engine vs oracle on a live graph)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for s in 3 4 5 6; do echo "== seed $s, 500 steps"; timeout -s KILL 300 python tools/fuzz_gpu.py --seed $s --steps 500 2>&1 | grep -v amdgpu.ids | tail -1; done
for s in 7 8; do echo "== seed $s, 300 steps, bursts of <= 800 updates, universe x20 (compactions)"; timeout -s KILL 400 python tools/fuzz_gpu.py --seed $s --steps 300 --burst 800 --universe 20 2>&1 | grep -v amdgpu.ids | tail -2; done
timeout -s KILL 300 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
