#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -5
timeout 200 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | tee $O/r02_22_batcher.txt
for B in 1 64 1024 4096 16384; do
  timeout 120 python bench.py --batch $B --no-cpu --configs off --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 batch $B: p50 host-id call ms', round(d['latency']['p50_batch_ms'],4), 'p95', round(d['latency']['p95_batch_ms'],4), 'device-resident ms', round(d['device_resident']['ms_per_batch'],4))"
done 2>&1 | tee $O/r02_22_small.txt
for U in 1 2 3; do
  echo "== C2 ACL_LOCAL_UPW=$U"; ACL_LOCAL_UPW=$U bash tools/levels.sh r02_c2_upw$U --workload C2 2>&1 | grep "last levels"
done 2>&1 | tee $O/r02_22_c2_upw.txt
