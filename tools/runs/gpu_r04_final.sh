#!/bin/bash
# round 4, final: the whole GPU suite, smoke(), the driver-style default bench (-> profiles/r04_bench.json), rocprofv3 --kernel-trace --stats of the
# SAME command, then stats + FETCH / WRITE passes per configuration (C4, C5-size replica, C3, C2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r04_final_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r04_final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout -s KILL 900 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err ) 2>&1 | grep real
python - <<P
import json
d=json.loads(open('$O/r04_bench.json').read().strip().splitlines()[-1])
print('DEFAULT value %.1f M/s (long run %.1f) p50 %.4f kernel %.1f us frac %.3f parity %s string %.1f C2 %.1f C3 %.0f traffic %s' % (d['value']/1e6, d['host_ids']['long_run']['decisions_per_s']/1e6, d['latency']['p50_batch_ms'], d['roofline'].get('kernel_avg_us', 0), d['roofline']['frac'], d.get('parity'), d['string_path']['decisions_per_s']/1e6 if 'decisions_per_s' in d.get('string_path', {}) else -1, d['configs']['C2']['value']/1e6, d['configs']['C3']['value'], d['roofline'].get('traffic')))
P
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/r04_bench_stats -o r -- python $R/bench.py --steps 5 --no-cpu > $O/prof_r04_bench_stats.log 2>&1
cd $R
bash tools/prof_c4.sh r04f_c4 > /dev/null 2>&1
bash tools/prof_c4.sh r04f_c5r --workload C5 --replica > /dev/null 2>&1
bash tools/prof_c4.sh r04f_c3 --workload C3 > /dev/null 2>&1
bash tools/prof_c4.sh r04f_c2 --workload C2 > /dev/null 2>&1
ls $O/prof/
