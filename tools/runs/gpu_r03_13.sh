#!/bin/bash
# round 3, call 13: chained-pipeline lanes as the admission queue (callers 4..N wait for one of 3 contexts) + completer thread for tickets: tests, then host-id throughput by caller count / window
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 500 python -m pytest tests/test_callers_gpu.py tests/test_fullscale_gpu.py::test_c4_full_every_entry_point tests/test_write_path_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
one() {
  python bench.py --no-cpu --steps 60 --configs off $1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1'.ljust(34), 'host-id M/s', round(d['value']/1e6,1), '| device M/s', round(d['device_resident']['decisions_per_s']/1e6,1), '| single-call p50 ms', round(d['latency']['p50_batch_ms'],4), '| equal', d['host_ids']['answers_equal_device_leg'])"
}
( for c in 1 2 3 4 8 16; do one "--callers $c"; done
  for w in 1 2 3 4 6; do one "--pipeline submit --window $w"; done ) 2>&1 | tee $O/r03_13_hostid_modes.txt
