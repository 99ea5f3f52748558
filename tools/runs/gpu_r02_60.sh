#!/bin/bash
# micro-batcher: blocking callers (wake-up tree) and the completion-queue form, spin knobs A/B on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 200 python -m pytest tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
run() {
  echo "== $1" | tee -a $O/r02_60_batcher_ab.txt
  env $1 timeout 60 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | grep '"mode"' | grep -v "one device pass" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print('cq ' if 'logical_callers' in d else 'blk', d.get('logical_callers', d['threads']), round(d['checks_per_s']/1e3), 'k/s', 'lat_us', d['mean_latency_us'], 'items/pass', round(d['checks']/max(1,d['batcher_passes']),1), 'err', d['errors'])
" | tee -a $O/r02_60_batcher_ab.txt
}
rm -f $O/r02_60_batcher_ab.txt
run "X=default"
run "ACL_BATCHER_IDLE_SPIN_US=50 ACL_BATCHER_POLL_SPIN_US=50"
run "ACL_BATCHER_IDLE_SPIN_US=50 ACL_BATCHER_POLL_SPIN_US=50 BENCH_POLLERS=4"
run "ACL_BATCHER_IDLE_SPIN_US=100 ACL_BATCHER_POLL_SPIN_US=100 BENCH_POLLERS=12"
run "ACL_BATCHER_IDLE_SPIN_US=30 ACL_BATCHER_POLL_SPIN_US=0"
run "ACL_BATCHER_IDLE_SPIN_US=0 ACL_BATCHER_POLL_SPIN_US=50 BENCH_POLLERS=4"
