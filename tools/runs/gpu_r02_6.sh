#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
export ACL_SKIP_C5_FULL=1
timeout 120 python -m pytest tests/test_write_path_gpu.py -m gpu -q --tb=long -p no:cacheprovider -s -k compaction > $O/r02_6_compaction.log 2>&1; echo "compaction test rc=$?"
tail -30 $O/r02_6_compaction.log
timeout 240 python -m pytest tests/test_fullscale_gpu.py tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r02_6_tests.log 2>&1; echo "tests rc=$?"
tail -8 $O/r02_6_tests.log
for WN in 1 2; do
  timeout 100 python bench.py --no-cpu --configs off --steps 40 --window $WN 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('window $WN: pipelined M/s', round(d['value']/1e6,1), 'unpipelined p50 ms', round(d['latency']['p50_batch_ms'],4), 'device M/s', round(d['device_resident']['decisions_per_s']/1e6,1))"
done 2>&1 | tee $O/r02_6_window.txt
timeout 300 python tools/write_latency.py 2>&1 | tail -1 | tee $O/r02_6_write_latency.json
