#!/bin/bash
# round 4, call 14: a lone caller's host-mapped launch with SKEWED units (first block's unit larger, last one's smaller: the items cross PCIe in
# block order) -- single 262 144-item call p50 for skew 0 / 8 / 16 / 24 / 32 / 40 %, with the batch in one launch (split 1) and cut in two (split 2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for S in 1 2; do for K in 0 8 16 24 32 40; do
  ACL_HOST_SPLIT=$S ACL_HOST_SKEW_PCT=$K timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 --callers 1 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('split $S skew $K%%: one caller %.1f M/s | p50 single call %.4f ms | equal to the device leg: %s' % (d['value']/1e6, d['latency']['p50_batch_ms'], d['host_ids']['answers_equal_device_leg']))"
done; done
