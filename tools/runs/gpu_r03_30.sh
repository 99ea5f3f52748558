#!/bin/bash
# round 3, call 30: trace of the submit / wait pipeline at windows 2, 3, 4 (C4, 24 steps)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
for w in 2 3 4; do
  ACL_TRACE_PIPELINE=1 timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 24 --warmup 4 --pipeline submit --window $w 2>$O/r03_30_trace_w$w.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('window $w value', round(d['value']/1e6,1), 'M/s ms/step', round(d['ms_per_step'],4))"
  grep -c aclgpu-pipeline $O/r03_30_trace_w$w.txt
done
