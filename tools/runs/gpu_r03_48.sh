#!/bin/bash
# round 3, call 48: poll the stream before blocking in hipStreamSynchronize (ACL_SYNC_SPIN_US): single-call latency, throughput, small batches
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  timeout -s KILL 300 python bench.py --workload $1 --no-cpu --configs off --strings off --steps 40 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 spin_us=${ACL_SYNC_SPIN_US:-0}: value %.1f M/s | single call p50 %.4f ms p95 %.4f | device-resident p50 %.4f ms' % (d['value']/1e6, d['latency']['p50_batch_ms'], d['latency']['p95_batch_ms'], d['device_resident']['p50_batch_ms']))"
}
for sp in 0 1000 0 1000; do ACL_SYNC_SPIN_US=$sp run C4; done
for sp in 0 1000; do ACL_SYNC_SPIN_US=$sp run C2; done
for sp in 0 1000; do echo "batch sweep, spin_us=$sp:"; ACL_SYNC_SPIN_US=$sp timeout 120 python tools/batch_sweep.py 2>/dev/null | head -6; done
