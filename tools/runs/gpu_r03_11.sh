#!/bin/bash
# round 3, call 11: waves per block of k_check_local (= waves sharing one unit) 2 / 4 / 8 / 16, and half-size dynamic units, same-box A/B on C4 and C2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
one() {  # $1 = env, $2 = workload
  env $1 python bench.py --no-cpu --steps 40 --legs device --configs off --workload $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$2', '$1'.ljust(90), 'device-resident M/s', round(d['device_resident']['decisions_per_s']/1e6,1), 'kernel us', round(1e3*d['device_resident']['kernel_ms_per_batch'],1))"
}
L=$R/spicedb-kubeapi-proxy_amd/lib
( for w in C4 C2; do
  one "X=base4" $w
  one "ACLGPU_LIB=$L/libaclgpu_w8.so" $w
  one "ACLGPU_LIB=$L/libaclgpu_w16.so" $w
  one "ACLGPU_LIB=$L/libaclgpu_w2.so" $w
  one "ACL_LOCAL_STATIC_PCT=50 ACL_LOCAL_DYN_UNIT=64" $w
  one "ACLGPU_LIB=$L/libaclgpu_w8.so ACL_LOCAL_STATIC_PCT=50 ACL_LOCAL_DYN_UNIT=128" $w
  one "X=base4" $w
done ) 2>&1 | sed "s#$L/##" | tee $O/r03_11_waves_per_block_ab.txt
