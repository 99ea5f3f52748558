#!/bin/bash
# round 4, call 3: 8-byte frontier entries and the reverse walk that clears what it marked -- parity first, then the A/B (lib vs the
# 16-byte build libaclgpu_e16.so) on C4 / C2 / the C5-size replica, then stats + FETCH / WRITE passes of the new kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -4
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$1: value %.1f M/s | device %.1f M/s kernel %.1f us | p50 single call %.4f ms' % (d['value']/1e6, d['device_resident']['decisions_per_s']/1e6, 1e3*d['device_resident']['kernel_ms_per_batch'], d['latency']['p50_batch_ms']))"; }
for rep in 1 2; do
for L in libaclgpu.so libaclgpu_e16.so; do
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/$L timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 2>/dev/null | tail -1 | line "C4 $L"
done; done
for L in libaclgpu.so libaclgpu_e16.so; do
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/$L timeout 300 python bench.py --workload C2 --no-cpu --configs off --strings off --steps 40 2>/dev/null | tail -1 | line "C2 $L"
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/$L timeout 600 python bench.py --workload C5 --replica --no-cpu --configs off --strings off --steps 20 2>/dev/null | tail -1 | line "C5R $L"
done
bash tools/prof_c4.sh r04_c4 > /dev/null 2>&1; ls gpurun_out/prof/r04_c4/*/ | head
bash tools/prof_c4.sh r04_c5r --workload C5 --replica > /dev/null 2>&1
bash tools/prof_c4.sh r04_c3 --workload C3 > /dev/null 2>&1
bash tools/prof_c4.sh r04_c2 --workload C2 > /dev/null 2>&1
ls gpurun_out/prof/
