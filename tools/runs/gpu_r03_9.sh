#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
ACL_DEBUG_REBUILD=1 timeout 600 python tools/dual_write_latency.py > $O/r03_9_dual_write.json 2> $O/r03_9_dual_write.err; echo "dual write rc=$?"
grep aclgpu $O/r03_9_dual_write.err | head -60; cat $O/r03_9_dual_write.json
