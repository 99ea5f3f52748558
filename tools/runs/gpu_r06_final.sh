#!/bin/bash
# round 6, final: the whole GPU suite (with the slowest tests listed), smoke(), the driver-style default bench (-> gpurun_out/r06_final/bench.json), rocprofv3
# --kernel-trace --stats of the SAME command, the differential fuzz campaigns and the stress harness (one replica and three in-process replicas)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06_final
mkdir -p $O
( time timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=12 > $O/gpu_tests.txt 2>&1 ) 2>&1 | grep real; echo "tests rc=$?"
tail -3 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<P
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('DEFAULT value %.1f M/s (long run %.1f) kernel %.1f us frac %.3f traffic_frac %.3f  C5R %.1f us frac %.3f stream %.0f M/s + %.0f lookups/s  mismatches %s/%s' % (d['value']/1e6, r['long_run_decisions_per_s']/1e6, r['kernel_avg_us'], r['frac'], r['traffic_frac'] or 0, r['c5r_kernel_avg_us'], r['c5r_frac'], r['c5r_stream_decisions_per_s']/1e6, r['c5r_stream_lookups_per_s'], r['c5r_parity_mismatches'], r['c5r_stream_mismatches']))
P
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/r06_bench_stats -o r -- python $R/bench.py --steps 5 --no-cpu > $O/prof_bench_stats.log 2>&1 )
python tools/rocprof_summary.py r06_bench_stats "$(find $R/gpurun_out/prof/r06_bench_stats -name '*.db' | head -1)" --workload "bench.py --steps 5 --no-cpu (every leg)" --steps 5 --kernel k_check_local --items 262144 --out $O > /dev/null 2>&1
ls $O
{
echo "# tools/runs/gpu_r06_final.sh: the differential fuzz on a live graph on the round-6 code (marked-through reverse relations, strict lookups, the entry prefetch; seeds 81 / 83: one user's pairs through CheckBulkPermissions twice + the keep mask -- with cycles the calls stay forward, --acyclic they take the reverse walk after the depth sweep), then the stress harness"
run() { echo "== $*"; timeout 900 python tools/fuzz_gpu.py "$@" 2>&1 | tail -1 | cut -c1-700; }
run --seed 71 --steps 600 --recycle
run --seed 72 --steps 500 --recycle --compact-early
run --seed 73 --steps 500 --recycle --schema combine
run --seed 75 --steps 150 --recycle --burst 300 --universe 3
run --seed 78 --steps 600 --expiry
run --seed 81 --steps 400 --recycle
run --seed 83 --steps 500 --recycle --acyclic
} > $O/fuzz.txt 2>&1
tail -6 $O/fuzz.txt | cut -c1-300
g++ -O2 -std=c++17 tools/engine_stress.cpp -Iinclude -Lspicedb-kubeapi-proxy_amd/lib -laclgpu -lpthread -Wl,-rpath,$R/spicedb-kubeapi-proxy_amd/lib -o /tmp/engine_stress
{ echo "# tools/engine_stress 25 (every call shape of the seam at once -- string batches and PostFilter calls by reverse walk among them --, each answer compared with the same call made alone), one replica | ACL_DEVICES=0,0,0"; timeout 120 /tmp/engine_stress 25 2>&1 | tail -1; ACL_DEVICES=0,0,0 timeout 120 /tmp/engine_stress 25 2>&1 | tail -1; } > $O/engine_stress.txt 2>&1
cat $O/engine_stress.txt
python tools/keep_route_probe.py > $O/keep.txt 2>/dev/null
ACL_DEBUG_KEEP=1 python tools/keep_route_probe.py --calls 10 --sizes 65536 2>&1 | grep "keep route" | awk "NR%4==0" > $O/keep_trace.txt
python tools/string_cold_probe.py > $O/string_cold.txt 2>/dev/null
cat $O/keep.txt | cut -c1-150; cat $O/string_cold.txt
python tools/list_filter_probe.py --calls 10 > $O/list_filter.txt 2>/dev/null; cut -c1-260 $O/list_filter.txt | grep -E '^[0-9]'
