#!/usr/bin/env python3
"""LookupResources batch-size sweep on C3 (power users repeated to fill a batch): wall time per call and the kernel's HIP-event time.
usage: python tools/lookup_bench.py [sizes...]     env: ACL_REV_LOCAL=0 (level loop), ACL_REV_ROWS=device (rows through a device buffer)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64, 256, 1024]
w = workloads.c3()
with aclgpu.Engine(w.schema) as e:
    w.load(e)
    e.snapshot()
    rt, perm, st = w.check
    words = max(1, (e.object_count(rt) + 31) // 32)
    rng = np.random.default_rng(3)
    for kind in ("power", "ordinary"):
        for n in sizes:
            subs = np.resize(w.lookup_subjects, n).astype(np.uint32) if kind == "power" else rng.integers(0, 10000, size=n).astype(np.uint32)
            hb = e.host_alloc(n * words * 4 + n * 8)
            out = (hb[:n * words * 4].view(np.uint32).reshape(n, words), hb[n * words * 4:].view(np.uint64))
            for _ in range(3):
                e.lookup_ids_batch(rt, perm, st, "", subs, out=out)
            e.stats_reset()
            e.set_timing(True)
            lat = []
            reps = 30 if n <= 256 else 8
            for _ in range(reps):
                t0 = time.perf_counter()
                _b, cnt = e.lookup_ids_batch(rt, perm, st, "", subs, out=out)
                lat.append(time.perf_counter() - t0)
            e.set_timing(False)
            s = e.stats()
            kms = s["rev_local_ms"] if s["rev_local_passes"] else s["kernel_ms"]
            print(f"{kind:8s} n={n:5d} wall p50 {1e6 * np.median(lat):8.1f} us  kernel {1e3 * kms / reps:8.1f} us  lookups/s {n / np.median(lat):10.0f}  ids/lookup {float(cnt.mean()):8.1f} "
                  f"kernel={'k_rev_local' if s['rev_local_passes'] else 'level loop'}", flush=True)
            e.host_free(hb)
