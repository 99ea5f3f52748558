#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
L=$R/spicedb-kubeapi-proxy_amd/lib
export ACL_SKIP_C5_FULL=1
timeout 300 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_list_filter.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
echo "== new"; timeout 200 python tools/string_path.py 2>&1 | tail -1
echo "== prev"; ACLGPU_LIB=$L/libaclgpu_prev.so timeout 200 python tools/string_path.py 2>&1 | tail -1
for LIB in "" "$L/libaclgpu_prev.so"; do
  ACLGPU_LIB=$LIB timeout 120 python bench.py --no-cpu --configs off --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench string leg ($LIB): M/s', round(d['string_path']['decisions_per_s']/1e6,1), d['string_path']['ms_per_batch'])"
done
