#!/bin/bash
# round 3, call 43: large host batches read / answered across PCIe by the kernel itself (ACL_HOSTMAP_MAX) vs copied: single-call latency and throughput
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  timeout -s KILL 300 python bench.py --workload $1 --no-cpu --configs off --strings off --steps 30 --callers $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 callers=$2 hostmap_max=${ACL_HOSTMAP_MAX:-8192}: value %.1f M/s | single call p50 %.4f ms p95 %.4f | kernel %.1f us | equal %s' % (d['value']/1e6, d['latency']['p50_batch_ms'], d['latency']['p95_batch_ms'], d['roofline']['kernel_avg_us'] or 0, d['host_ids']['answers_equal_device_leg']))"
}
for hm in 8192 300000; do for c in 1 2; do ACL_HOSTMAP_MAX=$hm run C4 $c; done; done
for hm in 8192 300000; do ACL_HOSTMAP_MAX=$hm run C2 1; done
