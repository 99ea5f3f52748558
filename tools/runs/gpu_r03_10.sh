#!/bin/bash
# round 3, call 10: k_check_local with static units for part of the batch + small hand-out units for the rest (DESIGN 8.2's untried experiment), same-box A/B on C4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
one() {
  env $1 python bench.py --no-cpu --steps 40 --legs device --configs off 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1'.ljust(60), 'device-resident M/s', round(d['device_resident']['decisions_per_s']/1e6,1), 'kernel us', round(1e3*d['device_resident']['kernel_ms_per_batch'],1))"
}
( one "ACL_LOCAL_STATIC_PCT=100"
  one "ACL_LOCAL_STATIC_PCT=85 ACL_LOCAL_DYN_UNIT=32"
  one "ACL_LOCAL_STATIC_PCT=75 ACL_LOCAL_DYN_UNIT=32"
  one "ACL_LOCAL_STATIC_PCT=75 ACL_LOCAL_DYN_UNIT=16"
  one "ACL_LOCAL_STATIC_PCT=75 ACL_LOCAL_DYN_UNIT=64"
  one "ACL_LOCAL_STATIC_PCT=60 ACL_LOCAL_DYN_UNIT=32"
  one "ACL_LOCAL_STATIC_PCT=50 ACL_LOCAL_DYN_UNIT=32"
  one "ACL_LOCAL_STATIC_PCT=50 ACL_LOCAL_DYN_UNIT=64"
  one "ACL_LOCAL_STATIC_PCT=100" ) 2>&1 | tee $O/r03_10_static_dynamic_ab.txt
