#!/bin/bash
# round 4, call 22: the string path's process-to-process swing (0.31 - 0.53 ms at 65 536 items) against the number of interning threads -- the boxes
# show 256 CPUs behind a 16-core cgroup quota (cpu.max = 1600000 100000)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cat /sys/fs/cgroup/cpu.max
for rep in 1 2 3 4; do for T in 32 16 12; do
  echo -n "threads $T: "; ACL_INTERN_THREADS=$T timeout 600 python tools/string_shapes.py 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(' | '.join('%s %.3f ms' % (k, v['p50_ms']) for k, v in d.items()))"
done; done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
