#!/bin/bash
# round 3, call 47: the default bench line as the driver runs it (three callers), twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
for i in 1 2; do
( time timeout -s KILL 900 python bench.py > $O/r03_47_bench_$i.json 2> $O/r03_47_bench_$i.err ) 2>&1 | grep real
python - <<P
import json
d=json.loads(open('$O/r03_47_bench_$i.json').read().strip().splitlines()[-1])
print('value %.1f M/s ms/step %.4f p50 %.4f kernel %.1f us frac %.3f traffic %.0f parity %s | string %.1f | C2 %.1f | C3 %.0f | cpu %.0f/s x%d' % (d['value']/1e6, d['ms_per_step'], d['p50_batch_ms'], d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['roofline']['traffic'], d['parity'], d['string_path']['decisions_per_s']/1e6, d['configs']['C2']['value']/1e6, d['configs']['C3']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']))
print(d['config']['timed'])
P
done
