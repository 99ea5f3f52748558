#!/usr/bin/env python3
"""usage (GPU box): [ACL_DEBUG_KEEP=1] python tools/keep_route_probe.py [--calls 40] -- PostFilter's shape on C4's named graph (K list items x one template for ONE
user): acl_check_bulk_keep_v / _packed through the reverse-walk route and acl_check_bulk_v of the same pairs (CheckBulkPermissions itself: the pair form of the route --
on C4's recursive schema after the depth sweep, engine.cpp no_object_is_deep; ACL_KEEP_ROUTE_MIN=0 or ACL_DEPTH_SWEEP=0 in the environment gives the forward path); one line per (K, user).  The masks are
compared with the id path's answers.  ACL_DEBUG_KEEP=1 prints the route's phase times per call on stderr."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd"), os.path.join(ROOT, "tests")]
import aclgpu  # noqa: E402
import bench  # noqa: E402
from aclgpu import workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=40)
ap.add_argument("--sizes", default="1024,16384,65536")
ap.add_argument("--workload", default="c4", choices=["c4", "c3"], help="c3: the 1 M-relationship cluster -> namespace -> pod graph with 64 power users -- a schema without recursion, "
                "on which CheckBulkPermissions itself (acl_check_bulk_v of one user's pairs) takes the reverse walk too")
a = ap.parse_args()
w = workloads.c4() if a.workload == "c4" else workloads.c3(batch=65536)
eng = aclgpu.Engine(w.schema, contexts=3, eager_contexts=True)
bench.name_objects(eng, w)
w.load(eng)
eng.snapshot()
rt, perm_name, st = w.check
names = w.names
grants = sorted({int(x) for x in w.res[:65536:2731]})[:24]
eng.write([(aclgpu.OP_TOUCH, (rt, names[rt][g], "viewer", st, "user-sparse", "")) for g in grants])
out = {}
for m in [int(x) for x in a.sizes.split(",")]:
    m = min(m, len(w.res))
    users = (("batch_user", int(w.subj[0])), ("other_user", int(w.subj[m // 2])), ("sparse_user", None))
    if a.workload == "c3":
        users = (("power_user", int(w.lookup_subjects[0])), ("ordinary_user", 3), ("sparse_user", None))
    for who, u in users:
        uname = "user-sparse" if u is None else names[st][u]
        qk = [(rt, names[rt][int(r)], perm_name, st, uname, "") for r in w.res[:m]]
        off = np.arange(m + 1, dtype=np.uint32)
        if u is None:
            want = np.isin(w.res[:m], np.asarray(grants, dtype=w.res.dtype))
        else:
            tp, te = eng.check_bulk_ids(eng.make_items(rt, perm_name, w.res[:m], st, "", np.full(m, u, dtype=np.uint32)))
            want = (tp == 2) & (te == 0)
        row = {"kept": int(want.sum())}
        def pairs_views(prep, _off):  # CheckBulkPermissions of the same pairs: a keep mask out of its permissionships and errors
            p_, e_ = eng.check_bulk_views(prep)
            return (p_ == 2) & (e_ == 0)
        for form, call, prep in (("keep_v", eng.check_bulk_keep_views, eng.make_check_views(qk)), ("keep_packed", eng.check_bulk_keep_packed, eng.make_check_packed(qk)),
                                 ("check_bulk_v", pairs_views, eng.make_check_views(qk))):
            before = eng.stats()["keep_route_calls"]
            ok = bool(np.array_equal(call(prep, off).astype(bool), want))
            ts = []
            for _ in range(a.calls):
                t1 = time.perf_counter()
                call(prep, off)
                ts.append(time.perf_counter() - t1)
            row[form] = {"M_items_per_s": round(m / float(np.mean(ts)) / 1e6, 1), "p50_ms": round(1e3 * float(np.median(ts)), 4), "best_ms": round(1e3 * min(ts), 4),
                         "mask_ok": ok, "by_reverse_walk": int(eng.stats()["keep_route_calls"] - before)}
        if a.workload == "c4" and u is not None:
            # K items x TWO templates (pod#view and pod#creator for the same user): one walk per template under one evaluation (engine.cpp keep_by_reverse_walks);
            # beside it the same 2 K pairs by id through the forward walk + the AND
            q2 = [x for r in w.res[:m] for x in ((rt, names[rt][int(r)], perm_name, st, uname, ""), (rt, names[rt][int(r)], "creator", st, uname, ""))]
            off2 = np.arange(0, 2 * m + 1, 2, dtype=np.uint32)
            cp, ce = eng.check_bulk_ids(eng.make_items(rt, "creator", w.res[:m], st, "", np.full(m, u, dtype=np.uint32)))
            want2 = want & (cp == 2) & (ce == 0)
            prep2 = eng.make_check_views(q2)
            before = eng.stats()["keep_route_calls"]
            ok2 = bool(np.array_equal(eng.check_bulk_keep_views(prep2, off2).astype(bool), want2))
            ts = []
            for _ in range(a.calls):
                t1 = time.perf_counter()
                eng.check_bulk_keep_views(prep2, off2)
                ts.append(time.perf_counter() - t1)
            it2 = np.empty(2 * m, dtype=aclgpu.ITEM_DTYPE)
            it2[0::2] = eng.make_items(rt, perm_name, w.res[:m], st, "", np.full(m, u, dtype=np.uint32))
            it2[1::2] = eng.make_items(rt, "creator", w.res[:m], st, "", np.full(m, u, dtype=np.uint32))
            tf = []
            for _ in range(a.calls):
                t1 = time.perf_counter()
                eng.check_bulk_ids(it2)
                tf.append(time.perf_counter() - t1)
            row["keep_v_two_templates"] = {"M_items_per_s": round(m / float(np.mean(ts)) / 1e6, 1), "p50_ms": round(1e3 * float(np.median(ts)), 4), "mask_ok": ok2, "kept": int(want2.sum()),
                                           "walks": int(eng.stats()["keep_route_calls"] - before), "forward_by_id_p50_ms": round(1e3 * float(np.median(tf)), 4)}
        out[f"{m}/{who}"] = row
        print(m, who, json.dumps(row), flush=True)
eng.close()
