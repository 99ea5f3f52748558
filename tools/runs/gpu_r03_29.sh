#!/bin/bash
# round 3, call 29: cache-line name slots + 32 interning threads for large batches -- the whole GPU suite, then the string leg three times
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_29_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 $O/r03_29_tests.log
run() {
  timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --steps 10 2>$O/r03_29_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
sp=d['string_path']['sizes']
print(' | '.join(k+' views mean %.1f p50 %.1f  c_str mean %.1f p50 %.1f M/s' % (v['views']['decisions_per_s']/1e6, int(k)/v['views']['p50_ms']/1e3, v['c_strings']['decisions_per_s']/1e6, int(k)/v['c_strings']['p50_ms']/1e3) for k,v in sp.items()), '| eq', all(v['views']['answers_equal_id_path'] and v['c_strings']['answers_equal_id_path'] for v in sp.values()))"
}
for i in 1 2 3; do run; done
