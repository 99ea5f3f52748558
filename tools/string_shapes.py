#!/usr/bin/env python3
"""usage (GPU box): tools/string_shapes.py -- the string entry point (acl_check_bulk_v) on the proxy's own batch shapes, on the C4 graph with NAMED objects:
  random     every item another pod and another user (what bench.py's string leg measures)
  one user   65 536 pods for ONE user: a PostFilter call (postfilter.go:88-119)
  3 per pod  21 845 pods x 3 permissions-worth of pairs for one user: F = 3 templates per list item, pairs of one item adjacent
Answers are compared with the id path's.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

w = workloads.c4()
n = 65536
out = {}
pod = lambda i: f"ns{i % 997}/pod-{i}"  # noqa: E731
usr = lambda i: f"user-{i % 997}-{i}"  # noqa: E731
with aclgpu.Engine(w.schema) as e:
    # names for every pod and user BEFORE the numeric bulk load (as bench.py name_objects): ids follow interning order, so name k <-> id k
    for t, fn in (("pod", pod), ("user", usr)):
        last = -1
        for i in range(w.nobjects[t]):
            last = e.intern(t, fn(i))
        assert last == w.nobjects[t] - 1
    w.load(e)
    rng = np.random.default_rng(1)
    shapes = {
        "random": [("pod", pod(int(r)), "view", "user", usr(int(s)), "") for r, s in zip(w.res[:n], w.subj[:n])],
        "one user": [("pod", pod(int(r)), "view", "user", usr(int(w.subj[0])), "") for r in w.res[:n]],
        "3 per pod": [("pod", pod(int(w.res[k // 3])), "view", "user", usr(int(w.subj[0])), "") for k in range(n)],
    }
    for name, qs in shapes.items():
        prep = e.make_check_views(qs)
        ids = e.make_items("pod", "view", np.array([e.find("pod", q[1]) for q in qs], dtype=np.uint32), "user", "", np.array([e.find("user", q[4]) for q in qs], dtype=np.uint32))
        wp, we = e.check_bulk_ids(ids)
        for _ in range(5):
            p, er = e.check_bulk_views(prep)
        assert np.array_equal(p, wp) and np.array_equal(er, we), name
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            e.check_bulk_views(prep)
            ts.append(time.perf_counter() - t0)
        out[name] = {"items": n, "p50_ms": round(1e3 * float(np.median(ts)), 4), "decisions_per_s": round(n / float(np.median(ts)), 1), "equal_to_id_path": True}
print(json.dumps(out))
