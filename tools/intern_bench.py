#!/usr/bin/env python3
"""usage: tools/intern_bench.py [items] [pods] [users] -- names -> ids alone (acl_resolve_bulk_v on a store-only engine: no GPU needed).

The host half of a string call: `pods` + `users` names in the tables (C4's 845 000 + 100 000 by default), `items` random (pod, user) pairs whose names
lie scattered in one blob, as a cgo shim's Go strings would.  Prints ms per call (median / best of 30); ACL_INTERN_THREADS=k sets the pool's size, fewer than 4 096 items stay on the calling thread."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
npods = int(sys.argv[2]) if len(sys.argv) > 2 else 845000
nusers = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
SCHEMA = "definition user {}\ndefinition pod {\n  relation viewer: user\n  permission view = viewer\n}\n"
e = aclgpu.Engine(SCHEMA, store_only=True)
t0 = time.time()
pods = [f"ns-{i % 977:03d}/pod-{i:07d}" for i in range(npods)]
users = [f"team-{i % 113:03d}|user-{i:06d}" for i in range(nusers)]
for p in pods:
    e.intern("pod", p)
for u in users:
    e.intern("user", u)
rng = np.random.default_rng(5)
qs = [("pod", pods[int(a)], "view", "user", users[int(b)], "") for a, b in zip(rng.integers(0, npods, m), rng.integers(0, nusers, m))]
pv = e.make_check_views(qs)
items, err = e.resolve_bulk_views(pv)
assert not err.any()
assert all(int(items["resource_id"][i]) == e.find("pod", qs[i][1]) and int(items["subject_id"][i]) == e.find("user", qs[i][4]) for i in range(0, m, 997))
ts = []
for _ in range(30):
    t1 = time.perf_counter()
    e.resolve_bulk_views(pv)
    ts.append(time.perf_counter() - t1)
env = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ACL_"))
print(f"{m} items, tables {npods} + {nusers} names, {env or 'defaults'}: median {1e3 * np.median(ts):.3f} ms best {1e3 * min(ts):.3f} ms "
      f"= {m / np.median(ts) / 1e6:.1f} M items/s (setup {time.time() - t0:.0f} s)")
e.close()
