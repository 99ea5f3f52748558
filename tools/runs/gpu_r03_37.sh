#!/bin/bash
# round 3, call 37: the default bench line (what the driver runs), twice; then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
for i in 1 2; do
  ( time timeout -s KILL 900 python bench.py > $O/r03_37_bench_$i.json 2> $O/r03_37_bench_$i.err ) 2>&1 | grep real
  python - <<P
import json
d=json.loads(open('$O/r03_37_bench_$i.json').read().strip().splitlines()[-1])
print('value %.1f M/s  ms/step %.4f  p50 %.4f  kernel %.1f us  frac %.3f  parity %s  cpu %s' % (d['value']/1e6, d['ms_per_step'], d['p50_batch_ms'], d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['parity'], {k:d['cpu_baseline'][k] for k in ('value','cores','kind')}))
print(' string 65536: %.1f M/s  C2 %.1f M/s  C3 %.0f lookups/s  single_checks %s' % (d['string_path']['decisions_per_s']/1e6, d['configs']['C2']['value']/1e6, d['configs']['C3']['value'], json.dumps(d.get('single_checks'))[:300]))
P
done
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_37_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r03_37_tests.log
