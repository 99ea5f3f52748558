#!/bin/bash
# round 3, call 26: string path A/B -- slices on/off, intern threads 16/32/48
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
run() {
  timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --steps 20 2>$O/r03_26_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
sp=d['string_path']['sizes']
print('slices=$ACL_STRING_SLICES threads=$ACL_INTERN_THREADS | ' + ' | '.join(k+' views %.1f c_str %.1f M/s (p50 %.3f ms)' % (v['views']['decisions_per_s']/1e6, v['c_strings']['decisions_per_s']/1e6, v['views']['p50_ms']) for k,v in sp.items()), '| eq', all(v['views']['answers_equal_id_path'] and v['c_strings']['answers_equal_id_path'] for v in sp.values()))"
}
for s in 0 1 0 1; do ACL_STRING_SLICES=$s run; done
for t in 32 48; do for s in 0 1; do ACL_INTERN_THREADS=$t ACL_STRING_SLICES=$s run; done; done
