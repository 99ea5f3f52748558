#!/bin/bash
# round 3, call 44: host-mapped large batches under 2 ... 16 callers, kernels one at a time (token) or free
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  timeout -s KILL 300 python bench.py --workload $1 --no-cpu --configs off --strings off --steps 40 --callers $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 callers=$2 hostmap_max=${ACL_HOSTMAP_MAX:-8192} token=${ACL_HOSTMAP_TOKEN:-1}: value %.1f M/s | single call p50 %.4f ms | equal %s' % (d['value']/1e6, d['latency']['p50_batch_ms'], d['host_ids']['answers_equal_device_leg']))"
}
export ACL_HOSTMAP_MAX=300000
for t in 1 0; do for c in 2 4 8 16; do ACL_HOSTMAP_TOKEN=$t run C4 $c; done; done
for t in 1 0; do for c in 2 4; do ACL_HOSTMAP_TOKEN=$t run C2 $c; done; done
