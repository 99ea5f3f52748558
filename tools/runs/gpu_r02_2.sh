#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
export ACL_SKIP_C5_FULL=1
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_list_filter.py tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/r02_2_tests.log 2>&1; echo "tests rc=$?"
tail -15 $O/r02_2_tests.log
timeout 900 bash tools/ab.sh 2>&1 | tee $O/r02_2_ab.txt
timeout 300 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | tee $O/r02_2_batcher.txt
for WN in 1 2 3 4; do
  timeout 120 python bench.py --no-cpu --configs off --steps 40 --window $WN 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('window $WN: pipelined M/s', round(d['value']/1e6,1), 'unpipelined p50 ms', round(d['latency']['p50_batch_ms'],4), 'device M/s', round(d['device_resident']['decisions_per_s']/1e6,1))"
done 2>&1 | tee $O/r02_2_window.txt
