#!/bin/bash
# usage (GPU box): tools/levels.sh <tag> [bench args] -- kernel-trace only; prints the last batch's per-level k_expand durations
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o r -- python $R/bench.py --steps 4 --warmup 2 --no-cpu --legs device --configs off "$@" > $O/run.log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("$O/*.db")[0]
con=sqlite3.connect(db)
rows=con.execute("select name,start,end-start from kernels order by start").fetchall()
ex=[(n,d) for n,s,d in rows if 'k_expand' in n]
L=int("${LEVELS:-6}")
print("last levels us:", [round(d/1e3,1) for n,d in ex[-L:]], "sum", round(sum(d for n,d in ex[-L:])/1e3,1), "| k_check_local us:", [round(d/1e3,1) for n,s,d in rows if 'k_check_local' in n][-3:])
PY
tail -1 $O/run.log | cut -c1-200
