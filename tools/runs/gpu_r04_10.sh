#!/bin/bash
# round 4, call 10: is the walk sensitive to level quantisation?  device-resident throughput at 131 072 / 262 144 / 524 288 items
# (units of 256 / 512 / 1 024 requests per 16-wave block)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for B in 131072 262144 524288; do
  timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 30 --batch $B 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('batch $B: host ids %.1f M/s | device %.1f M/s | kernel %.1f us = %.3f ns per item | p50 single call %.4f ms' % (d['value']/1e6, d['device_resident']['decisions_per_s']/1e6, 1e3*d['device_resident']['kernel_ms_per_batch'], 1e6*d['device_resident']['kernel_ms_per_batch']/$B, d['latency']['p50_batch_ms']))"
done
