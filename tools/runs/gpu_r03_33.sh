#!/bin/bash
# round 3, call 33: lanes created + copy paths warmed by the first chip-filling batch: no ~7 ms hipMemcpyAsync stall later
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACL_TRACE_PIPELINE=1 timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 24 --warmup 4 --pipeline submit --window 3 2>$O/r03_33_trace_w3.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('window 3 value', round(d['value']/1e6,1), 'M/s ms/step', round(d['ms_per_step'],4))"
python - <<'P'
import collections
ev=collections.defaultdict(dict)
for line in open('gpurun_out/r03_33_trace_w3.txt'):
    if line.startswith('aclgpu-pipeline'):
        _,i,what,t=line.split(); ev[int(i)][what]=float(t)
for i in sorted(ev):
    e=ev[i]
    print(i, 'begin->context %.0f us, h2d enqueue %.0f us, submit->finished %.0f us' % (e['context']-e.get('stage_begin',e.get('stage_begin_blocking')), e['h2d_enqueued']-e['before_h2d'], e['finished']-e['submit']))
P
run() {
  timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$*: value', round(d['value']/1e6,1), 'M/s ms/step', round(d['ms_per_step'],4))"
}
for c in 3 4 5 16; do run --pipeline blocking --callers $c; done
run --pipeline blocking --callers 4
