#!/usr/bin/env python3
"""Write -> fully-consistent read latency on the full-size C4 graph (10 M relationships): a kube-style create
(2 relationships naming new objects) followed by one CheckPermission, patched snapshot vs forced rebuild."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu
from aclgpu import workloads

w = workloads.c4()
e = aclgpu.Engine(w.schema)
w.load(e)
e.snapshot()
e.lookup("pod", "view", "user", "nobody")  # builds + uploads the reverse rows once
lat, wl, lk = [], [], []
for i in range(200):
    t0 = time.perf_counter()
    e.write([(aclgpu.OP_CREATE, ("pod", f"ns/p{i}", "creator", "user", f"paul{i}", "")), (aclgpu.OP_TOUCH, ("pod", f"ns/p{i}", "namespace", "namespace", "ns", ""))])
    t1 = time.perf_counter()
    assert e.check("pod", f"ns/p{i}", "view", "user", f"paul{i}") == (2, 0)
    lat.append(time.perf_counter() - t1)
    wl.append(t1 - t0)
    if i % 4 == 0:  # a list request right after a write: LookupResources must see it too
        e.write([(aclgpu.OP_TOUCH, ("pod", f"ns/q{i}", "creator", "user", f"paul{i}", ""))])
        t2 = time.perf_counter()
        assert e.lookup("pod", "view", "user", f"paul{i}") == {f"ns/p{i}", f"ns/q{i}"}
        lk.append(time.perf_counter() - t2)
st = e.stats()
# forced rebuild: a bulk load bypasses the change feed
reb = []
for i in range(3):
    e.add_edges("pod", "creator", "user", "", np.array([i], dtype=np.uint32), np.array([i], dtype=np.uint32))
    t1 = time.perf_counter()
    e.check("pod", "ns/p0", "view", "user", "paul0")
    reb.append(time.perf_counter() - t1)
print(json.dumps({"workload": "C4 10M relationships", "writes": 200, "write_ms_p50": 1e3 * float(np.median(wl)),
                  "check_after_write_ms_p50": 1e3 * float(np.median(lat)), "check_after_write_ms_p95": 1e3 * float(np.percentile(lat, 95)),
                  "lookup_after_write_ms_p50": 1e3 * float(np.median(lk)), "snapshot_patches": st["snapshot_patches"], "snapshot_builds": st["snapshot_builds"],
                  "check_after_forced_rebuild_ms": [round(1e3 * x, 1) for x in reb]}))
