#!/bin/bash
# round 3, call 42: intern pool pinned to the caller's NUMA node or not -- the string leg, 5 processes each (the spread is between processes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
lscpu | grep -E "Socket|NUMA node|Model name" | head -8
run() {
  timeout -s KILL 200 python bench.py --workload C4 --no-cpu --configs off --steps 4 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
v=d['string_path']['sizes']['65536']['views']; c=d['string_path']['sizes']['65536']['c_strings']; s=d['string_path']['sizes']['16384']['views']
print('pin=$ACL_INTERN_PIN 65536 views mean %.3f p50 %.3f ms (%.0f M/s) | c_str p50 %.3f | 16384 views p50 %.3f ms (%.0f M/s)' % (v['ms_per_batch'], v['p50_ms'], 65.536/v['ms_per_batch'], c['p50_ms'], s['p50_ms'], 16.384/s['ms_per_batch']))"
}
for i in 1 2 3 4 5; do ACL_INTERN_PIN=0 run; ACL_INTERN_PIN=1 run; done
