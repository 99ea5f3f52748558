#!/usr/bin/env python3
"""usage (GPU box): [ACL_DEBUG_INTERN_PIECES=1] python tools/string_cold_probe.py -- acl_check_bulk_v calls of 1 024 / 16 384 / 65 536 random (pod, user) pairs on a
named graph, issued (a) back to back -- the interning pool's workers are still polling when the next call arrives -- and (b) 2 ms apart: every worker asleep, as
a proxy's calls mostly find them.  ms per call: median / best of 30."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402

SCHEMA = """definition user {}
definition namespace { relation viewer: user
 permission view = viewer }
definition pod { relation namespace: namespace
 relation viewer: user
 permission view = viewer + namespace->view }"""
rng = np.random.default_rng(7)
NU, NNS, NPOD = 100_000, 2_000, 845_000
e = aclgpu.Engine(SCHEMA, eager_contexts=True)
pods = [f"ns{int(rng.integers(0, NNS))}/pod-{p}" for p in range(NPOD)]
for p in pods:
    e.intern("pod", p)
for u in range(NU):
    e.intern("user", f"user-{u}")
rels = [f"namespace:ns{n}#viewer@user:user-{int(u)}" for n in range(NNS) for u in rng.integers(0, NU, size=10)]
rels += [f"pod:{pods[i]}#namespace@namespace:{pods[i].split('/')[0]}" for i in range(0, NPOD, 9)]
rels = list(dict.fromkeys(rels))
for i in range(0, len(rels), 1000):
    e.write([(aclgpu.OP_TOUCH, r) for r in rels[i:i + 1000]])
out = {}
for n in (1024, 16384, 65536):
    qs = [("pod", pods[int(i)], "view", "user", f"user-{int(u)}", "") for i, u in zip(rng.integers(0, NPOD, n), rng.integers(0, NU, n))]
    pv = e.make_check_views(qs)
    for _ in range(3):
        e.check_bulk_views(pv)
    row = {}
    for mode, gap in (("back_to_back", 0.0), ("2ms_apart", 0.002)):
        ts = []
        for _ in range(30):
            if gap:
                time.sleep(gap)
            t1 = time.perf_counter()
            e.check_bulk_views(pv)
            ts.append(time.perf_counter() - t1)
        row[mode] = {"median_ms": round(1e3 * float(np.median(ts)), 4), "best_ms": round(1e3 * min(ts), 4), "M_per_s_at_median": round(n / float(np.median(ts)) / 1e6, 1)}
    out[n] = row
    print(n, json.dumps(row), flush=True)
e.close()
