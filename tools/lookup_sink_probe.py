#!/usr/bin/env python3
"""usage (GPU box): [ACL_REV_SINK=0] python tools/lookup_sink_probe.py [--workload c4|c5r] -- LookupResources whose RESULT slot other permissions expand, on the benchmark
graph: pod#creator (feeds pod#view), namespace#view (feeds pod#view through the arrow), namespace#viewer; beside pod#view (nobody expands it).  p50 of 40 single lookups per
target and subject; run once as it is and once with ACL_REV_SINK=0 (the walk goes on beyond the result slot, and over a big type it cannot defer its heavy rows)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c4", choices=["c4", "c5r"])
a = ap.parse_args()
w = workloads.c4() if a.workload == "c4" else workloads.c5()
eng = aclgpu.Engine(w.schema, contexts=2, eager_contexts=True)
w.load(eng)
eng.snapshot()
rt, perm, st = w.check
subs = [int(w.subj[0]), int(w.subj[len(w.subj) // 2]), int(w.subj[7])]
out = {"sink": os.environ.get("ACL_REV_SINK", "1") != "0"}
for t, p in ((rt, perm), (rt, "creator"), ("namespace", "view"), ("namespace", "viewer")):
    row = []
    for u in subs:
        bm, cnt = eng.lookup_ids_batch(t, p, st, "", [u])
        keep = (bm, cnt)
        ts = []
        for _ in range(40):
            t1 = time.perf_counter()
            eng.lookup_ids_batch(t, p, st, "", [u], out=keep)
            ts.append(time.perf_counter() - t1)
        row.append({"allowed": int(cnt[0]), "p50_us": round(1e6 * float(np.median(ts)), 1), "crc": int(np.bitwise_xor.reduce(bm[0]))})
    out[f"{t}#{p}"] = row
print(json.dumps(out))
