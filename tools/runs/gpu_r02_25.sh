#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
L=$R/spicedb-kubeapi-proxy_amd/lib
export ACL_SKIP_C5_FULL=1
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py tests/test_callers_gpu.py tests/test_write_path_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/r02_25_tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/r02_25_tests.log
run() { # name env...
  echo "== $*"
  env "$@" bash tools/levels.sh r02_25_x 2>&1 | grep "last levels"
  env "$@" bash tools/levels.sh r02_25_y --workload C2 2>&1 | grep "last levels"
}
run A=default
run ACLGPU_LIB=$L/libaclgpu_w3.so
run ACL_LOCAL_UPW=2
for B in 1 64 1024 4096 16384 65536; do
  timeout 120 python bench.py --batch $B --no-cpu --configs off --steps 50 --legs device 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 batch $B: device-resident ms', round(d['device_resident']['ms_per_batch'],4), 'kernel ms', round(d['device_resident']['kernel_ms_per_batch'],4))"
done 2>&1 | tee $O/r02_25_small.txt
