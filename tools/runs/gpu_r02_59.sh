#!/bin/bash
# micro-batcher wake-up tree: A/B of the knobs on one box (engine_callers.cpp), after the callers' GPU tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 200 python -m pytest tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
run() {
  echo "== $1" | tee -a $O/r02_59_batcher_ab.txt
  env $1 timeout 60 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | grep micro-batched | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print(d['threads'], round(d['checks_per_s']/1e3), 'k/s', 'lat_us', d['mean_latency_us'], 'passes', d['batcher_passes'], 'items/pass', round(d['checks']/max(1,d['batcher_passes']),1), 'err', d['errors'])
" | tee -a $O/r02_59_batcher_ab.txt
}
rm -f $O/r02_59_batcher_ab.txt
nproc | tee -a $O/r02_59_batcher_ab.txt; cat /sys/fs/cgroup/cpu.max | tee -a $O/r02_59_batcher_ab.txt
run "ACL_BATCHER_CHAIN=0"
run "ACL_BATCHER_CHAIN=1"
run "ACL_BATCHER_CHAIN=1 ACL_BATCHER_QUEUES=16"
run "ACL_BATCHER_CHAIN=1 ACL_BATCHER_FANOUT=3"
run "ACL_BATCHER_CHAIN=1 ACL_BATCHER_PER_QUEUE=12"
run "ACL_BATCHER_CHAIN=1 ACL_BATCHER_DISPATCHERS=2"
run "ACL_BATCHER_CHAIN=1 ACL_BATCHER_SPINNERS=0"
run "ACL_BATCHER_CHAIN=1 ACL_BATCHER_SPINNERS=12"
run "ACL_BATCHER_CHAIN=0"
run "ACL_BATCHER_CHAIN=1"
