#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ACL_SKIP_C5_FULL=1
timeout 500 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_sharded_gpu.py tests/test_write_path_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
for i in 1 2; do timeout 120 python bench.py --workload C3 --no-cpu --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 lookups/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['kernel_ms_per_step'],4), 'launches', d['rev_expand_launches_per_step'], 'single p50 ms', round(d['p50_single_lookup_ms'],4))"; done
