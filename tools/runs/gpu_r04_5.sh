#!/bin/bash
# round 4, call 5: the device-state refactor (in-process replicas), LDS answers, NOMARK seed ops, API validation: the whole GPU suite, then
# the one-process bench with 1 / 2 / 4 logical replicas on this GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -12
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1: n_gpus %d value %.1f M/s | long run %.1f M/s | replica calls %s' % (d['n_gpus'], d['value']/1e6, d['host_ids']['long_run']['decisions_per_s']/1e6, d['host_ids']['replica_calls']))"; }
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 2>&1 | tail -1 | line "C4 one replica"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 --devices 0,0 2>&1 | tail -1 | line "C4 --devices 0,0"
timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 --devices 0,0,0,0 --callers 2 2>&1 | tail -1 | line "C4 --devices 0,0,0,0"
