#!/usr/bin/env python3
"""usage (GPU box): python tools/string_path.py -- decisions/s of acl_check_bulk (5 C strings per item interned on the host, then one
device pass) on a graph whose objects HAVE names (bench.py's string leg uses numeric names that resolve to nothing: all misses)."""
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
 relation creator: user
 permission view = viewer + creator }
definition pod { relation namespace: namespace
 relation viewer: user
 relation creator: user
 permission view = viewer + creator + namespace->view }"""
rng = np.random.default_rng(7)
NU, NNS, NPOD = 50_000, 2_000, 200_000
e = aclgpu.Engine(SCHEMA)
t0 = time.time()
rels = []
for p in range(NPOD):
    ns = int(rng.integers(0, NNS))
    rels.append(f"pod:ns{ns}/pod-{p}#namespace@namespace:ns{ns}")
    rels.append(f"pod:ns{ns}/pod-{p}#creator@user:user-{int(rng.integers(0, NU))}")
for n in range(NNS):
    for u in rng.integers(0, NU, size=20):
        rels.append(f"namespace:ns{n}#viewer@user:user-{int(u)}")
rels = list(dict.fromkeys(rels))
for i in range(0, len(rels), 1000):
    e.write([(aclgpu.OP_TOUCH, r) for r in rels[i:i + 1000]])
t_load = time.time() - t0
pods = [r.split("#")[0][4:] for r in rels if r.startswith("pod:") and "#namespace@" in r]
out = {"graph": f"{len(rels)} named relationships, {NPOD} pods, {NU} users", "load_s": round(t_load, 1)}
for n in (1, 64, 1024, 16384, 65536):
    idx = rng.integers(0, NPOD, size=n)
    us = rng.integers(0, NU, size=n)
    qs = [("pod", pods[i], "view", "user", f"user-{int(u)}", "") for i, u in zip(idx, us)]
    prepared = e.make_check_strings_named(qs)
    want = e.check_bulk(qs)
    for _ in range(3):
        got = e.check_bulk_prepared(prepared)
    assert got[0].tolist() == want[0] and got[1].tolist() == want[1]
    ts = []
    for _ in range(15):
        t1 = time.perf_counter()
        e.check_bulk_prepared(prepared)
        ts.append(time.perf_counter() - t1)
    best = min(ts)
    out[f"items_{n}"] = {"ms": round(1e3 * best, 4), "decisions_per_s": round(n / best), "has_fraction": round(float((got[0] == 2).mean()), 3)}
print(json.dumps(out))
e.close()
