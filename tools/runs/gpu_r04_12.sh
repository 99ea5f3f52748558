#!/bin/bash
# round 4, call 12: the reference's dual-write sequence on the 10 M graph across background compactions WITH id recycling at work
# (quarantine shortened to 1 s for the run: 37 k kube writes take ~20 s), then with the production quarantine
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
ACL_ID_QUARANTINE_MS=1000 timeout 900 python tools/dual_write_latency.py 2>/dev/null | tail -1 > gpurun_out/r04_dual_write_q1s.json; tail -c 1500 gpurun_out/r04_dual_write_q1s.json; echo
timeout 900 python tools/dual_write_latency.py 2>/dev/null | tail -1 > gpurun_out/r04_dual_write.json; tail -c 1500 gpurun_out/r04_dual_write.json; echo
