#!/bin/bash
# usage: tools/pmc.sh <tag> "<COUNTER ...>" [bench args]   -- one rocprofv3 --pmc pass (own run, --kernel-trace only)
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout ${PMC_TIMEOUT:-90} rocprofv3 --kernel-trace --pmc $CTRS -d $O -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --legs device --configs off "$@" > $O/run.log 2>&1
tail -2 $O/run.log | cut -c1-300
