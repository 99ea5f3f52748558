#!/usr/bin/env python3
"""usage (GPU box): python tools/small_call_probe.py -- what a small acl_check_bulk_ids call is made of: wall p50 of the call, the kernel's own duration
(HIP events on the context's stream; timing switches the spin-wait off, so the wall figure is taken in a separate pass) for n = 1, 8, 64, 1024 on C2's graph."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

w = workloads.c2(scale=0.5)
e = aclgpu.Engine(w.schema)
w.load(e)
rt, perm, st = w.check
items = e.make_items(rt, perm, w.res, st, "", w.subj)
e.check_bulk_ids(items[:64])
for n in (1, 8, 64, 1024):
    it = items[:n].copy()
    for _ in range(50):
        e.check_bulk_ids(it)
    ts = []
    for _ in range(400):
        t0 = time.perf_counter()
        e.check_bulk_ids(it)
        ts.append(time.perf_counter() - t0)
    e.set_timing(True)
    e.stats_reset()
    for _ in range(200):
        e.check_bulk_ids(it)
    st_ = e.stats()
    e.set_timing(False)
    print(f"n={n:5d}: call p50 {1e6 * np.median(ts):6.1f} us (python + ctypes included) | kernel {1e3 * st_['local_ms'] / max(1, st_['local_passes']):6.2f} us per launch ({st_['local_passes']} launches)")
e.close()
