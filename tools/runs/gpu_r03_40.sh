#!/bin/bash
# round 3, call 40: fuzz with write bursts on a larger universe (compactions), then the fuzz test
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for s in 7 8; do echo "== seed $s, 200 steps, bursts of <= 800 updates, universe x20"; timeout -s KILL 150 python tools/fuzz_gpu.py --seed $s --steps 200 --burst 800 --universe 20 2>&1 | grep -v amdgpu.ids | tail -2; done
timeout -s KILL 200 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
