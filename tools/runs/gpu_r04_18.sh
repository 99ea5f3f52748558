#!/bin/bash
# round 4, call 18: soak -- tools/engine_stress (every call shape of the seam at once, each answer compared with the same call made alone) for
# 25 s on one replica, on three logical replicas, and with the host-batch slices on four streams
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
[ -x tools/bin/engine_stress ] || g++ -O2 -std=c++17 tools/engine_stress.cpp -I include -L spicedb-kubeapi-proxy_amd/lib -laclgpu -lpthread -Wl,-rpath,'$ORIGIN/../../spicedb-kubeapi-proxy_amd/lib' -o tools/bin/engine_stress
timeout 120 tools/bin/engine_stress 25 2>&1 | tail -2
ACL_DEVICES=0,0,0 timeout 120 tools/bin/engine_stress 25 2>&1 | tail -2
ACL_HOST_SPLIT=4 timeout 120 tools/bin/engine_stress 25 2>&1 | tail -2
