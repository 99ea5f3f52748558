#!/bin/bash
# round 3, call 41: the whole GPU suite (incl. the fuzz test), smoke(), the default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider --durations=8 > $O/r03_41_tests.log 2>&1; echo "tests rc=$?"
tail -14 $O/r03_41_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout -s KILL 900 python bench.py > $O/r03_41_bench.json 2> $O/r03_41_bench.err ) 2>&1 | grep real
python - <<P
import json
d=json.loads(open('$O/r03_41_bench.json').read().strip().splitlines()[-1])
print('value %.1f M/s kernel %.1f us frac %.3f traffic %s (%s)' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_detail']['source'][:30]))
print('parity', d['parity'], 'string', round(d['string_path']['decisions_per_s']/1e6,1), 'C2', round(d['configs']['C2']['value']/1e6,1), 'C3', round(d['configs']['C3']['value']))
P
