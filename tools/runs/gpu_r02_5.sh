#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
export ACL_SKIP_C5_FULL=1
timeout 900 python -m pytest tests/test_write_path_gpu.py tests/test_fullscale_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s > $O/r02_5_tests.log 2>&1; echo "tests rc=$?"
grep -E "worst read-after-write|passed|failed|Error" $O/r02_5_tests.log | tail -8
for WN in 1 2 3; do
  timeout 120 python bench.py --no-cpu --configs off --steps 40 --window $WN 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('window $WN: pipelined M/s', round(d['value']/1e6,1), 'unpipelined p50 ms', round(d['latency']['p50_batch_ms'],4), 'device M/s', round(d['device_resident']['decisions_per_s']/1e6,1))"
done 2>&1 | tee $O/r02_5_window.txt
timeout 900 python tools/write_latency.py 2>&1 | tail -1 | tee $O/r02_5_write_latency.json
