#!/bin/bash
# round 4, call 7: a lone caller's host batch as concurrent slices on streams of their own (ACL_HOST_SPLIT = 1 / 2 / 3 / 4):
# single 262 144-item call p50 and one-caller throughput at 262 144 and 1 048 576 items; three callers for the regression check
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1: value %.1f M/s | long run %.1f M/s | p50 single call %.4f ms' % (d['value']/1e6, d['host_ids']['long_run']['decisions_per_s']/1e6, d['latency']['p50_batch_ms']))"; }
for S in 1 2 3 4; do
  export ACL_HOST_SPLIT=$S
  timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 --callers 1 2>&1 | tail -1 | line "split $S: C4 262144 x 1 caller"
  timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 20 --batch 1048576 --callers 1 2>&1 | tail -1 | line "split $S: C4 1048576 x 1 caller"
done
for S in 1 2; do
  export ACL_HOST_SPLIT=$S
  timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 2>&1 | tail -1 | line "split $S: C4 262144 x 3 callers"
done
