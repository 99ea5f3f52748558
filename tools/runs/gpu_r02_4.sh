#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
export ACL_SKIP_C5_FULL=1
timeout 1200 python -m pytest tests/test_write_path_gpu.py tests/test_callers_gpu.py tests/test_engine_gpu.py tests/test_fullscale_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s > $O/r02_4_tests.log 2>&1; echo "tests rc=$?"
grep -E "worst read-after-write|passed|failed|Error" $O/r02_4_tests.log | tail -8
timeout 600 python bench.py > $O/r02_4_bench.json 2> $O/r02_4_bench.err; echo "bench rc=$?"; tail -2 $O/r02_4_bench.err
python - <<PY
import json
try:
    d=json.loads(open("$O/r02_4_bench.json").read().strip().splitlines()[-1])
    print("value M/s", round(d["value"]/1e6,1), "device M/s", round(d["device_resident"]["decisions_per_s"]/1e6,1), "p50 ms", round(d["p50_batch_ms"],4), "roofline", round(d["roofline"]["frac"],4), round(d["roofline"]["kernel_avg_us"],1), "parity", d.get("parity"))
    for k,v in d.get("configs",{}).items():
        print(k, {x:v.get(x) for x in ("value","parity","error")} if isinstance(v,dict) else v)
except Exception as e: print("bench parse failed", e)
PY
timeout 300 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | tee $O/r02_4_batcher.txt
for B in 1 64 1024 4096; do
  timeout 120 python bench.py --batch $B --no-cpu --configs off --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 batch $B: p50 host-id call ms', round(d['latency']['p50_batch_ms'],4), 'p95', round(d['latency']['p95_batch_ms'],4), 'device-resident ms', round(d['device_resident']['ms_per_batch'],4))"
done 2>&1 | tee $O/r02_4_small.txt
timeout 600 python tools/write_latency.py 2>&1 | tail -1 | tee $O/r02_4_write_latency.json
