#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_engine_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
for WN in 1 2 3; do
  timeout 100 python bench.py --no-cpu --configs off --steps 40 --pipeline submit --window $WN 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('submit window $WN: host-id M/s', round(d['value']/1e6,1), d['host_ids'].get('answers_equal_device_leg'))"
done
timeout 100 python bench.py --no-cpu --configs off --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocking 2 callers: host-id M/s', round(d['value']/1e6,1))"
