#!/bin/bash
# round 4, call 8: the concurrent-slices policy as shipped (two streams for batches beyond one launch; a lone caller's 262 144-item batch cut in
# two): parity test, then one / three callers at 262 144 and 1 048 576 items
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -k "concurrent_slices or overflow or sub_batched" 2>&1 | tail -4
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1: value %.1f M/s | long run %.1f M/s | p50 single call %.4f ms' % (d['value']/1e6, d['host_ids']['long_run']['decisions_per_s']/1e6, d['latency']['p50_batch_ms']))"; }
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 --callers 1 2>&1 | tail -1 | line "C4 262144 x 1 caller"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 2>&1 | tail -1 | line "C4 262144 x 3 callers"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 20 --batch 1048576 --callers 1 2>&1 | tail -1 | line "C4 1048576 x 1 caller"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 20 --batch 1048576 2>&1 | tail -1 | line "C4 1048576 x 3 callers"
ACL_HOST_SPLIT=1 timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 20 --batch 1048576 2>&1 | tail -1 | line "split off: C4 1048576 x 3 callers"
