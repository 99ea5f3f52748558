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
# ---- background compaction: creates of NEW pods until half of the pod tables' headroom is used (that starts a background
# build) and on past the point where round 1 had to rebuild synchronously; one Check right after every write
npod = w.nobjects["pod"]
need = int(npod * 0.25 * 0.75)  # headroom = 25 % + 1024 rows; the build starts when 90 % of the rows are taken
per = 500
clat = []
st_a = e.stats()
for b in range(0, need, per):
    ups = []
    for k in range(per):
        ups.append((aclgpu.OP_TOUCH, ("pod", f"cmp/p{b + k}", "creator", "user", f"paul{k % 200}", "")))
        ups.append((aclgpu.OP_TOUCH, ("pod", f"cmp/p{b + k}", "namespace", "namespace", "ns", "")))
    e.write(ups)
    t1 = time.perf_counter()
    assert e.check("pod", f"cmp/p{b + per - 1}", "view", "user", f"paul{(per - 1) % 200}") == (2, 0)
    clat.append(time.perf_counter() - t1)
    for _ in range(3):  # a few more reads between writes (the adoption happens on a read)
        t1 = time.perf_counter()
        e.check("pod", f"cmp/p{b}", "view", "user", "paul0")
        clat.append(time.perf_counter() - t1)
st_b = e.stats()
compaction = {"creates": need, "updates_per_write": 2 * per, "reads": len(clat), "read_ms_p50": 1e3 * float(np.median(clat)),
              "read_ms_p99": 1e3 * float(np.percentile(clat, 99)), "read_ms_max": 1e3 * float(np.max(clat)),
              "snapshot_compactions": st_b["snapshot_compactions"] - st_a["snapshot_compactions"],
              "synchronous_rebuilds": st_b["snapshot_builds"] - st_a["snapshot_builds"], "patches": st_b["snapshot_patches"] - st_a["snapshot_patches"],
              "note": "each write carries 1 000 updates (the per-write maximum, spicedb.go:35), so the read after it patches 1 000 relationships"}
print(json.dumps({"workload": "C4 10M relationships", "writes": 200, "background_compaction": compaction, "write_ms_p50": 1e3 * float(np.median(wl)),
                  "check_after_write_ms_p50": 1e3 * float(np.median(lat)), "check_after_write_ms_p95": 1e3 * float(np.percentile(lat, 95)),
                  "lookup_after_write_ms_p50": 1e3 * float(np.median(lk)), "snapshot_patches": st["snapshot_patches"], "snapshot_builds": st["snapshot_builds"],
                  "check_after_forced_rebuild_ms": [round(1e3 * x, 1) for x in reb]}))
