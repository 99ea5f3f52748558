#!/bin/bash
# round 4, call 19: host-id throughput against the number of blocking callers (the bench's default is 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for C in 3 3 3; do
  timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 20 --callers $C 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('callers $C: value %.1f M/s | long run %.1f M/s' % (d['value']/1e6, d['host_ids']['long_run']['decisions_per_s']/1e6))"
done
