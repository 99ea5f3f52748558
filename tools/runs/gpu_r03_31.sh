#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
for w in 3; do
  ACL_TRACE_PIPELINE=1 timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 24 --warmup 4 --pipeline submit --window $w 2>$O/r03_31_trace_w$w.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('window $w value', round(d['value']/1e6,1), 'M/s ms/step', round(d['ms_per_step'],4))"
  grep "aclgpu-pipeline [0-7] " $O/r03_31_trace_w$w.txt | awk '{printf "%s:%s@%s  ", $2,$3,$4} END{print ""}' | fold -w 220
done
