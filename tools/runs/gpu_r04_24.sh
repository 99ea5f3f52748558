#!/bin/bash
# round 4, call 24: the differential fuzz once more on the kernel with the push-time recheck (C4 schema: the monotone instantiations are the ones it changed)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "== $*"; timeout 900 python tools/fuzz_gpu.py "$@" 2>&1 | tail -1 | cut -c1-500; }
run --seed 61 --steps 700
run --seed 62 --steps 500 --recycle --compact-early
run --seed 63 --steps 200 --burst 300 --universe 3
timeout 600 python -m pytest tests/test_fullscale_gpu.py tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
