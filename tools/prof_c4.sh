#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_c4.sh <tag> [bench args...]
# rocprofv3 kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes of bench.py; raw dbs land in gpurun_out/prof/<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --legs device --configs off "$@" > $O/stats.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --legs device --configs off "$@" > $O/fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --legs device --configs off "$@" > $O/write.log 2>&1
grep -h '^{' $O/stats.log | tail -1
