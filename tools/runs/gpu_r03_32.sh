#!/bin/bash
# round 3, call 32: submit / wait windows 2, 3, 4, 6 and blocking callers 2, 4, 8 with a warm-up that reaches the steady state
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
run() {
  timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$*: value', round(d['value']/1e6,1), 'M/s ms/step', round(d['ms_per_step'],4))"
}
for w in 2 3 4 6; do run --pipeline submit --window $w; done
for c in 2 4 8; do run --pipeline blocking --callers $c; done
