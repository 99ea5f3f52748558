#!/bin/bash
# round 4, call 13: the walk's sensitivity to the number of bucket gathers -- one MORE 16-byte gather per child in flush_simple (same row, answers
# unchanged): what one gather per child costs is what a hashed-row format with one-gather lookups could save at most.  `third5`: only every
# fifth lane issues it -- is the cost per active lane (a mostly-masked second gather would pay) or per instruction (it would not)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/spicedb-kubeapi-proxy_amd/lib
for V in libaclgpu libaclgpu_third libaclgpu_third5 libaclgpu libaclgpu_third libaclgpu_third5; do
  ACLGPU_LIB=$L/$V.so timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 30 --legs device 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$V: device %.1f M/s | kernel %.1f us | HAS fraction %.4f' % (d['device_resident']['decisions_per_s']/1e6, 1e3*d['device_resident']['kernel_ms_per_batch'], d.get('has_fraction', -1)))"
done
