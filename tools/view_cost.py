#!/usr/bin/env python3
"""usage: [ACLGPU_LIB=...] python tools/view_cost.py [expiring relationships = 300000] -- what Store::view() (the copy a background snapshot build reads)
costs with many expiring relationships in the store (no GPU needed: a store-only engine and the compaction self-check's phase 0)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
os.environ["ACL_DEBUG_REBUILD"] = "1"
import aclgpu  # noqa: E402
from tests import kat_runner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
b = kat_runner.load_bootstrap()
e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
now = 1_700_000_000
e.set_now(now)
t0 = time.time()
for i in range(0, n, 1000):
    e.write([(aclgpu.OP_CREATE, ("workflow", f"w{j}", "idempotency_key", "activity", f"a{j}", ""), now + 3600) for j in range(i, min(n, i + 1000))])
print(f"{n} expiring relationships written in {time.time() - t0:.1f} s", flush=True)
e.selfcheck_snapshot()
for _ in range(3):
    e.write([(aclgpu.OP_CREATE, ("workflow", f"x{time.time_ns()}", "idempotency_key", "activity", "z", ""), now + 3600)])
    e.selfcheck_compaction(0)  # phase 0: view() (timed on stderr) + a build from it
    assert e.selfcheck_compaction(1) in (True, False, None, 0, 1)
e.close()
