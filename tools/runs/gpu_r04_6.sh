#!/bin/bash
# round 4, call 6: after retiring the copying pipeline (host-mapped sub-passes), id recycling, watch wait / recheck, bootstrap yaml:
# the whole GPU suite; C4 with 262 144- and 1 048 576-item batches (the latter = 2 sub-passes per call); submit/wait windows
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1: value %.1f M/s | long run %.1f M/s | device %.1f M/s kernel %.1f us | p50 single call %.4f ms' % (d['value']/1e6, d['host_ids']['long_run']['decisions_per_s']/1e6, d['device_resident']['decisions_per_s']/1e6, 1e3*d['device_resident']['kernel_ms_per_batch'], d['latency']['p50_batch_ms']))"; }
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 2>&1 | tail -1 | line "C4 262144 x 3 callers"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 --callers 1 2>&1 | tail -1 | line "C4 262144 x 1 caller"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 20 --batch 1048576 --callers 1 2>&1 | tail -1 | line "C4 1048576 x 1 caller"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 20 --batch 1048576 2>&1 | tail -1 | line "C4 1048576 x 3 callers"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 --pipeline submit --window 3 2>&1 | tail -1 | line "C4 submit window 3"
