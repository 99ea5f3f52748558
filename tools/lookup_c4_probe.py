#!/usr/bin/env python3
"""usage (GPU box): [ACL_REV_LDS_ROWS=0] python tools/lookup_c4_probe.py -- single LookupResources(pod, view, user:U) on C4 (845 000 pods: a 106 KB row, which fits the
block's LDS): wall p50 per call and ids returned, for a few of the batch's users -- the rows-in-LDS walk against the byte-map + deferred-rows walk that types
beyond the LDS take (ACL_REV_LDS_ROWS=0 forces it here)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

w = workloads.c4()
with aclgpu.Engine(w.schema) as e:
    w.load(e)
    e.snapshot()
    rt, perm, st = w.check
    words = max(1, (e.object_count(rt) + 31) // 32)
    hb = e.host_alloc(words * 4 + 8)
    out = (hb[:words * 4].view(np.uint32).reshape(1, words), hb[words * 4:].view(np.uint64))
    for u in [int(x) for x in w.subj[:4]] + [int(w.nobjects["user"]) + 5]:
        for _ in range(10):
            e.lookup_ids_batch(rt, perm, st, "", [u], out=out)
        lat = []
        for _ in range(100):
            t0 = time.perf_counter()
            _b, cnt = e.lookup_ids_batch(rt, perm, st, "", [u], out=out)
            lat.append(time.perf_counter() - t0)
        print(f"ACL_REV_LDS_ROWS={os.environ.get('ACL_REV_LDS_ROWS', 'default')} user {u:7d} ids {int(cnt[0]):7d} wall p50 {1e6 * np.median(lat):8.1f} us", flush=True)
