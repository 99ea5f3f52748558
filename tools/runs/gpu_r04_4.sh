#!/bin/bash
# round 4, call 4: answers (has / err) of the single-launch walk in LDS, reverse seed ops without visited bits, API validation -- parity, then
# the A/B: lib (8-byte entries + LDS answers) vs e8g (8-byte entries, global answers) vs e16 (round 3's layout), then FETCH / WRITE passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -4
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1: value %.1f M/s | device %.1f M/s kernel %.1f us | p50 single call %.4f ms' % (d['value']/1e6, d['device_resident']['decisions_per_s']/1e6, 1e3*d['device_resident']['kernel_ms_per_batch'], d['latency']['p50_batch_ms']))"; }
for rep in 1 2; do
for L in libaclgpu.so libaclgpu_e8g.so libaclgpu_e16.so; do
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/$L timeout 300 python bench.py --workload C4 --no-cpu --configs off --strings off --steps 40 2>/dev/null | tail -1 | line "C4 $L"
done; done
for L in libaclgpu.so libaclgpu_e8g.so; do
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/$L timeout 300 python bench.py --workload C2 --no-cpu --configs off --strings off --steps 40 2>/dev/null | tail -1 | line "C2 $L"
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/$L timeout 600 python bench.py --workload C5 --replica --no-cpu --configs off --strings off --steps 20 2>/dev/null | tail -1 | line "C5R $L"
done
timeout 300 python bench.py --workload C3 --no-cpu --configs off --strings off --steps 40 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C3: %.2f M lookups/s, kernel %.1f us, single lookup p50 %.1f us' % (d['value']/1e6, d['roofline']['kernel_avg_us'], 1e3*d['p50_single_lookup_ms']))"
for t in "r04b_c4" "r04b_c5r --workload C5 --replica" "r04b_c3 --workload C3" "r04b_c2 --workload C2"; do bash tools/prof_c4.sh $t > /dev/null 2>&1; done
ls gpurun_out/prof/
