#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_engine_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
for CL in 1 2 3 4; do
  timeout 100 python bench.py --no-cpu --configs off --steps 60 --callers $CL 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('--callers $CL: host-id M/s', round(d['value']/1e6,1), 'single-call p50 ms', round(d['latency']['p50_batch_ms'],4), 'device M/s', round(d['device_resident']['decisions_per_s']/1e6,1), d['host_ids'].get('answers_equal_device_leg'))"
done 2>&1 | tee $O/r02_35_modes.txt
timeout 100 python bench.py --workload C2 --no-cpu --configs off --steps 100 --callers 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 --callers 2: host-id M/s', round(d['value']/1e6,1), 'device M/s', round(d['device_resident']['decisions_per_s']/1e6,1))" | tee -a $O/r02_35_modes.txt
