#!/usr/bin/env python3
"""Check batch-size sweep on C4 (device-resident calls): kernel time by HIP events and wall time per call, for one setting of the engine's
A/B knobs (read at acl_open: ACL_LOCAL_WIDE_MIN, ACL_LOCAL_STATIC_PCT, ...).  usage: python tools/batch_sweep.py [sizes...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import torch  # noqa: E402

import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [64, 1024, 4096, 16384, 32768, 65536, 131072, 262144]
w = workloads.c4()
rt, perm, st = w.check
with aclgpu.Engine(w.schema) as e:
    w.load(e)
    e.snapshot()
    items = e.make_items(rt, perm, w.res, st, "", w.subj)
    for n in sizes:
        d_items = torch.from_numpy(items[:n].view(np.uint8).copy()).cuda()
        d_perm = torch.zeros(n, dtype=torch.uint8, device="cuda")
        d_err = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(5):
            e.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
        e.stats_reset()
        e.set_timing(True)
        reps = 40
        lat = []
        for _ in range(reps):
            t0 = time.perf_counter()
            e.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
            lat.append(time.perf_counter() - t0)
        e.set_timing(False)
        s = e.stats()
        print(f"n={n:7d} kernel {1e3 * s['local_ms'] / max(1, s['local_passes']):8.1f} us  wall p50 {1e6 * np.median(lat):8.1f} us  M decisions/s {n / np.median(lat) / 1e6:8.1f}  "
              f"has {float((d_perm == 2).float().mean()):.3f}", flush=True)
