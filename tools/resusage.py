#!/usr/bin/env python3
"""usage: tools/resusage.py [-DNAME=VALUE ...] -- VGPRs / spills / scratch / LDS of every kernel in kernels.hip
(hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU)"""
import os
import re
import subprocess
import sys

R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spicedb-kubeapi-proxy_amd")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *sys.argv[1:], "-Rpass-analysis=kernel-resource-usage",
                      "-c", f"{R}/csrc/kernels.hip", "-o", "/tmp/resusage.o"], capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"\bName: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("acl::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    g = lambda key: v.get(key, "?")  # noqa: E731
    print(f"{name:34s} vgpr {g('VGPRs'):>4} sgpr {g('SGPRs'):>4} vspill {g('VGPRs Spill'):>3} sspill {g('SGPRs Spill'):>3} scratch {g('ScratchSize [bytes/lane]'):>4} "
          f"occ {g('Occupancy [waves/SIMD]'):>2} lds {g('LDS Size [bytes/block]'):>6}")
