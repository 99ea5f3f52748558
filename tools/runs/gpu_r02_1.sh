#!/bin/bash
# round 2, GPU call 1: correctness of the refactored engine + new kernels, first numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
export ACL_SKIP_C5_FULL=1
nproc; python -c "import os;print('affinity',len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r02_1_tests.log 2>&1; echo "tests rc=$?"
tail -40 $O/r02_1_tests.log
timeout 600 python bench.py > $O/r02_1_bench.json 2> $O/r02_1_bench.err; echo "bench rc=$?"; tail -3 $O/r02_1_bench.err
python - <<PY
import json
try:
    d=json.loads(open("$O/r02_1_bench.json").read().strip().splitlines()[-1])
    print("value M/s", round(d["value"]/1e6,1), "device M/s", round(d["device_resident"]["decisions_per_s"]/1e6,1), "p50 ms", d["p50_batch_ms"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel_avg_us"], "parity", d.get("parity"))
    print("string", d.get("string_path"), "cpu", d.get("cpu_baseline",{}).get("value"))
    for k,v in d.get("configs",{}).items():
        print(k, v if not isinstance(v,dict) else {x:v.get(x) for x in ("value","parity","roofline","error")})
except Exception as e: print("bench parse failed", e)
PY
timeout 300 bash tools/ab.sh 2>&1 | tee $O/r02_1_ab.txt
timeout 300 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | tee $O/r02_1_batcher.txt
for B in 64 1024 4096 8192 16384; do
  for LM in 8192 0; do
    ACL_LOCAL_MAX=$LM timeout 120 python bench.py --batch $B --no-cpu --configs off --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $B local_max $LM: p50 host ms', round(d['latency']['p50_batch_ms'],4), 'device ms', round(d['device_resident']['ms_per_batch'],4), 'pipelined M/s', round(d['value']/1e6,2))"
  done
done 2>&1 | tee $O/r02_1_small.txt
