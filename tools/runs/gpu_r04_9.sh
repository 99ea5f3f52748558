#!/bin/bash
# round 4, call 9: where the E8 / LDS-answers walk's wave-time goes (phase marks), and two block-shape variants:
#   w3     = 12 waves per block, <= 84 VGPRs (6 waves / SIMD), three children per lane and step
#   wide12 = 12 waves per block, 64 VGPRs
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/spicedb-kubeapi-proxy_amd/lib
ACLGPU_LIB=$L/libaclgpu_phases.so timeout 300 python tools/phases.py C4 2>&1 | tail -14
for V in libaclgpu libaclgpu_w3 libaclgpu_wide12 libaclgpu; do
  ACLGPU_LIB=$L/$V.so timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 30 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$V: host ids %.1f M/s | device %.1f M/s | kernel %.1f us | p50 single call %.4f ms' % (d['value']/1e6, d['device_resident']['decisions_per_s']/1e6, 1e3*d['device_resident']['kernel_ms_per_batch'], d['latency']['p50_batch_ms']))"
done
