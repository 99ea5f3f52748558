#!/usr/bin/env python3
"""usage (GPU box): [ACL_SPIN_MAX=0] python tools/lookup_latency.py -- wall p50 of acl_lookup_resources_batch on C3 for 1 / 4 / 16 / 64 power users (timing
off: the small-batch completion word is what is being measured; tools/lookup_bench.py has the kernels' own durations)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

w = workloads.c3()
with aclgpu.Engine(w.schema) as e:
    w.load(e)
    e.snapshot()
    rt, perm, st = w.check
    words = max(1, (e.object_count(rt) + 31) // 32)
    for n in (1, 4, 16, 64):
        subs = np.resize(w.lookup_subjects, n).astype(np.uint32)
        hb = e.host_alloc(n * words * 4 + n * 8)
        out = (hb[:n * words * 4].view(np.uint32).reshape(n, words), hb[n * words * 4:].view(np.uint64))
        for _ in range(20):
            e.lookup_ids_batch(rt, perm, st, "", subs, out=out)
        lat = []
        for _ in range(300):
            t0 = time.perf_counter()
            _b, cnt = e.lookup_ids_batch(rt, perm, st, "", subs, out=out)
            lat.append(time.perf_counter() - t0)
        print(f"ACL_SPIN_MAX={os.environ.get('ACL_SPIN_MAX', 'default')} n={n:3d} wall p50 {1e6 * np.median(lat):6.1f} us p95 {1e6 * np.percentile(lat, 95):6.1f} us  ids/lookup {float(cnt.mean()):.0f}", flush=True)
        e.host_free(hb)
