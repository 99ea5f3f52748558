#!/bin/bash
# round 5, final: the whole GPU suite, smoke(), the driver-style default bench (-> gpurun_out/r05_bench.json), rocprofv3 --kernel-trace --stats of the
# SAME command, stats + FETCH / WRITE passes per configuration (C4, C5-size replica, C3, C2), the long fuzz campaigns and the stress harness
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r05_gpu_tests.txt 2>&1; echo "tests rc=$?"
tail -3 $O/r05_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench.json 2> $O/r05_bench.err ) 2>&1 | grep real
python - <<P
import json
d=json.loads(open('$O/r05_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('DEFAULT value %.1f M/s (long run %.1f) p50 %.4f kernel %.1f us frac %.3f traffic_frac %.3f in-leg %.1f us x%.2f parity %s string %.1f C2 %.1f C3 %.0f C5R %.1f us frac %.3f tf %.3f' % (d['value']/1e6, r['long_run_decisions_per_s']/1e6, d['latency']['p50_batch_ms'], r['kernel_avg_us'], r['frac'], r['traffic_frac'] or 0, r['kernel_us_in_leg'], r['overlap_factor'], d.get('parity'), d['string_path']['decisions_per_s']/1e6 if 'decisions_per_s' in d.get('string_path', {}) else -1, d['configs']['C2']['value']/1e6, d['configs']['C3']['value'], r['c5r_kernel_avg_us'], r['c5r_frac'], r['c5r_traffic_frac'] or 0))
print('single checks', json.dumps(d.get('single_checks', {}).get('small_batch_p50_us')), json.dumps(d.get('single_checks', {}).get('completion_queue')))
P
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/r05_bench_stats -o r -- python $R/bench.py --steps 5 --no-cpu > $O/prof_r05_bench_stats.log 2>&1
cd $R
bash tools/prof_c4.sh r05f_c4 > /dev/null 2>&1
bash tools/prof_c4.sh r05f_c5r --workload C5 --replica > /dev/null 2>&1
bash tools/prof_c4.sh r05f_c3 --workload C3 > /dev/null 2>&1
bash tools/prof_c4.sh r05f_c2 --workload C2 > /dev/null 2>&1
ls $O/prof/ | grep r05
{
echo "# tools/runs/gpu_r05_final.sh: the differential fuzz on a live graph on the round-5 kernels (seeded rows patched in place, direct task lists), then the stress harness"
run() { echo "== $*"; timeout 900 python tools/fuzz_gpu.py "$@" 2>&1 | tail -1 | cut -c1-700; }
run --seed 61 --steps 600 --recycle
run --seed 62 --steps 500 --recycle --compact-early
run --seed 63 --steps 500 --recycle --schema combine
run --seed 65 --steps 150 --recycle --burst 300 --universe 3
run --seed 68 --steps 600 --expiry
} > $O/r05_fuzz.txt 2>&1
tail -6 $O/r05_fuzz.txt | cut -c1-300
{ echo "# tools/bin/engine_stress 25 (every call shape of the seam at once, each answer compared with the same call made alone), one replica | ACL_DEVICES=0,0,0"; timeout 120 tools/bin/engine_stress 25 2>&1 | tail -1; ACL_DEVICES=0,0,0 timeout 120 tools/bin/engine_stress 25 2>&1 | tail -1; } > $O/r05_engine_stress.txt 2>&1
cat $O/r05_engine_stress.txt
