#!/bin/bash
# round 4, call 23: the deep levels' expansion looks at the request's answer byte (LDS) once more before it writes the children -- fewer junk
# entries, fewer segments at the next level?  A/B on C4, the C5-size replica and C2 (answers are compared with the oracle's by the bench)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/spicedb-kubeapi-proxy_amd/lib
for V in libaclgpu libaclgpu_recheck libaclgpu libaclgpu_recheck; do
  ACLGPU_LIB=$L/$V.so timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 30 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$V: host ids %.1f M/s (long %.1f) | device %.1f M/s | kernel %.1f us | HAS %.4f' % (d['value']/1e6, d['host_ids']['long_run']['decisions_per_s']/1e6, d['device_resident']['decisions_per_s']/1e6, 1e3*d['device_resident']['kernel_ms_per_batch'], d.get('has_fraction', -1)))"
done
for V in libaclgpu libaclgpu_recheck; do
  ACLGPU_LIB=$L/$V.so timeout 600 python bench.py --workload C5 --replica --no-cpu --configs off --strings off --steps 20 --legs device 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('C5R $V: kernel %.1f us' % (1e3*d['device_resident']['kernel_ms_per_batch']))"
done
