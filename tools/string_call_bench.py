#!/usr/bin/env python3
"""usage (GPU box): python tools/string_call_bench.py [items] -- one string call (acl_check_bulk_v) with the object names resolved on the device
(ACL_DEVICE_NAMES=1) against the same call with the names resolved by the host's interning threads (the default), same process, same tables (845 000 + 100 000
names as on C4), random (pod, user) pairs; the graph behind the names is small, so the walk is short and the difference is the names' alone.
ACL_DEBUG_STRING_TIMING=1 adds the engine's own split of each call to stderr."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
npods, nusers = 845000, 100000
SCHEMA = "definition user {}\ndefinition pod {\n  relation viewer: user\n  permission view = viewer\n}\n"
pods = [f"ns-{i % 977:03d}/pod-{i:07d}" for i in range(npods)]
users = [f"team-{i % 113:03d}|user-{i:06d}" for i in range(nusers)]
rng = np.random.default_rng(5)
pa, ua = rng.integers(0, npods, m), rng.integers(0, nusers, m)
qs = [("pod", pods[int(a)], "view", "user", users[int(b)], "") for a, b in zip(pa, ua)]
rels = "\n".join(f"pod:{pods[int(a)]}#viewer@user:{users[int(b)]}" for a, b in list(zip(pa, ua))[::2])  # every other pair is granted
res = {}
for mode in ("1", "0", "1", "0"):
    os.environ["ACL_DEVICE_NAMES"] = mode
    e = aclgpu.Engine(SCHEMA, rels)
    for p in pods:
        e.intern("pod", p)
    for u in users:
        e.intern("user", u)
    pv = e.make_check_views(qs)
    perm, err = e.check_bulk_views(pv)
    ans = (perm.copy(), err.copy())
    ts = []
    for _ in range(40):
        t1 = time.perf_counter()
        e.check_bulk_views(pv)
        ts.append(time.perf_counter() - t1)
    st = e.stats()
    print(f"{m} items, ACL_DEVICE_NAMES={mode}: median {1e3 * np.median(ts):.3f} ms mean {1e3 * np.mean(ts):.3f} best {1e3 * min(ts):.3f} ms = {m / np.median(ts) / 1e6:.1f} M decisions/s "
          f"(HAS {int((perm == 2).sum())}, errors {int((err != 0).sum())}, calls with names resolved on the device {st['device_name_calls']})", flush=True)
    if res:
        assert np.array_equal(res["ans"][0], ans[0]) and np.array_equal(res["ans"][1], ans[1]), "the two paths disagree"
    res["ans"] = ans
    e.close()
