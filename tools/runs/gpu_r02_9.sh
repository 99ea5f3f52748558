#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
for MODE in "--pipeline blocking --callers 1" "--pipeline blocking --callers 2" "--pipeline blocking --callers 3" "--pipeline submit --window 2"; do
  timeout 100 python bench.py --no-cpu --configs off --steps 40 $MODE 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$MODE: host-id M/s', round(d['value']/1e6,1), 'single-call p50 ms', round(d['latency']['p50_batch_ms'],4), 'device M/s', round(d['device_resident']['decisions_per_s']/1e6,1))"
done 2>&1 | tee $O/r02_9_modes.txt
# profiles of the final kernel: rocprofv3 stats + FETCH/WRITE passes, device-resident leg only (every k_expand launch is a sequential one)
timeout 400 bash tools/prof_c4.sh r02_c4_v11 2>&1 | tail -2
python tools/rocprof_summary.py r02_c4_v11 $(ls $O/prof/r02_c4_v11/stats/*.db | head -1) $(ls $O/prof/r02_c4_v11/fetch/*.db | head -1) $(ls $O/prof/r02_c4_v11/write/*.db | head -1) --workload C4 --steps 5 --out $O/profiles_r02 2>&1 | tail -3
# beyond-L3 data point: the C5-size graph (100 M relationships, ~0.7 GB snapshot) as ONE replica
timeout 600 python bench.py --workload C5 --replica --steps 10 --configs off > $O/r02_9_c5r_bench.json 2> $O/r02_9_c5r.err; echo "c5 replica bench rc=$?"; tail -c 1500 $O/r02_9_c5r_bench.json | head -c 1500; echo
timeout 700 bash tools/prof_c4.sh r02_c5r --workload C5 --replica 2>&1 | tail -2
python tools/rocprof_summary.py r02_c5r_replica $(ls $O/prof/r02_c5r/stats/*.db | head -1) $(ls $O/prof/r02_c5r/fetch/*.db | head -1) $(ls $O/prof/r02_c5r/write/*.db | head -1) --workload C5R --steps 5 --out $O/profiles_r02 2>&1 | tail -3
ls $O/profiles_r02
