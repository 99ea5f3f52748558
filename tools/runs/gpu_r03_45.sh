#!/bin/bash
# round 3, call 45: host batches answered by the kernel across PCIe by default -- the whole GPU suite, the caller sweep, the default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_45_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r03_45_tests.log
run() {
  timeout -s KILL 300 python bench.py --workload $1 --no-cpu --configs off --strings off --steps 40 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 $2 hostmap_max=${ACL_HOSTMAP_MAX:-default}: value %.1f M/s | single call p50 %.4f ms (pageable buffers %.4f) | device-resident %.1f M/s | equal %s' % (d['value']/1e6, d['latency']['p50_batch_ms'], d['latency'].get('pageable_buffers_p50_ms',0), d['device_resident']['decisions_per_s']/1e6, d['host_ids']['answers_equal_device_leg']))"
}
for c in 1 2 3 4 8 16; do run C4 "--callers $c"; done
for w in 2 4; do run C4 "--pipeline submit --window $w"; done
ACL_HOSTMAP_MAX=8192 run C4 "--callers 2"
for c in 1 2 4; do run C2 "--callers $c"; done
ACL_HOSTMAP_MAX=8192 run C2 "--callers 2"
( time timeout -s KILL 900 python bench.py > $O/r03_45_bench.json 2> $O/r03_45_bench.err ) 2>&1 | grep real
python - <<P
import json
d=json.loads(open('$O/r03_45_bench.json').read().strip().splitlines()[-1])
print('DEFAULT value %.1f M/s p50 %.4f kernel %.1f us frac %.3f parity %s string %.1f C2 %.1f C3 %.0f' % (d['value']/1e6, d['p50_batch_ms'], d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['parity'], d['string_path']['decisions_per_s']/1e6, d['configs']['C2']['value']/1e6, d['configs']['C3']['value']))
P
