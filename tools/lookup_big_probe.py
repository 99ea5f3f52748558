#!/usr/bin/env python3
"""usage (GPU box): python tools/lookup_big_probe.py [scale] -- where a LookupResources over a BIG type spends its time (BASELINE configs[4]'s graph as one replica:
8.45 M pods at scale 1.0, a result row of 1 MB that lives in HBM, not in the block's LDS).  Per subject kind -- a user nobody has a relationship with (the
walk is empty: what is left is the row's copy-out, count and clearing), a direct viewer of namespaces, deep-group members -- the kernel's time (HIP events
inside the engine), the call's wall time and the ids returned; one subject per call (the proxy's shape, pkg/authz/lookups.go:65) and all of them in one call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
t0 = time.time()
w = workloads.c5(scale=scale)
E = {(e[0], e[1], e[2]): (e[4], e[5]) for e in w.edges}
subs = {"nobody": int(w.nobjects["user"]) + 7, "ns-viewer": int(np.bincount(E[("namespace", "viewer", "user")][1]).argmax()),
        "deep-0": int(w.lookup_subjects[0]), "deep-1": int(w.lookup_subjects[1]), "deep-2": int(w.lookup_subjects[2])}
# ... and a power user who views 6 000 namespaces directly: ~1 M pods in one lookup ("a lookup returning >= 1 M ids")
n_pw = min(6000, w.nobjects["namespace"])
subs["power-1M"] = int(w.nobjects["user"]) + 11
w.edges.append(("namespace", "viewer", "user", "", np.arange(n_pw, dtype=np.uint32), np.full(n_pw, subs["power-1M"], dtype=np.uint32)))
e = aclgpu.Engine(w.schema)
w.load(e)
rt, perm, st = w.check
print(f"graph: {w.ntuples} relationships, {w.nobjects[rt]} {rt}s; gen+load {time.time() - t0:.0f} s", flush=True)
words = max(1, (e.object_count(rt) + 31) // 32)
hb = e.host_alloc(len(subs) * (words * 4 + 8))
bufs = (hb[:len(subs) * words * 4].view(np.uint32).reshape(len(subs), words), hb[len(subs) * words * 4:].view(np.uint64))
one = (bufs[0][:1], bufs[1][:1])
for name, s in subs.items():
    e.lookup_ids_batch(rt, perm, st, "", [s], out=one)  # warm (reverse rows are built on first use)
    e.set_timing(True)
    ks, ws = [], []
    for _ in range(12):
        e.stats_reset()
        t1 = time.perf_counter()
        _bm, cnt = e.lookup_ids_batch(rt, perm, st, "", [s], out=one)
        ws.append(time.perf_counter() - t1)
        stt = e.stats()
        ks.append(1e3 * (stt["rev_local_ms"] if stt["rev_local_passes"] else stt["expand_ms"]))
    e.set_timing(False)
    print(f"{name:10s} ids {int(cnt[0]):8d}  kernel us median {np.median(ks):8.1f} min {min(ks):8.1f}  call us median {1e6 * np.median(ws):8.1f}  "
          f"({'k_rev_local' if stt['rev_local_passes'] else 'level loop'}, row {words * 4} B)", flush=True)
e.set_timing(True)
e.stats_reset()
t1 = time.perf_counter()
_bm, cnts = e.lookup_ids_batch(rt, perm, st, "", list(subs.values()), out=bufs)
wall = time.perf_counter() - t1
stt = e.stats()
print(f"all {len(subs)} in one call: kernel us {1e3 * stt['rev_local_ms']:.1f}, call us {1e6 * wall:.1f}, ids {cnts.tolist()}")
e.close()
