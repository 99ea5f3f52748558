#!/usr/bin/env python3
"""usage (GPU box): [ACL_DEBUG_LIST=1] python tools/list_filter_probe.py [--calls 20] [--sizes 1000,10000,65536] [--workload c3|c4]
The LIST-level filters on the bytes of a kube list response: acl_filter_list_response (filterListResponse, postfilter.go:17-55:
decode, K x F checks, re-encode) and acl_prefilter_response (filterList, responsefilterer.go:376-400, over a LookupResources bitmap), K pods of the named
benchmark graph in a PodList of realistic item size, for a dense user and a sparse one.  Per line: ms per call, items/s, body MB/s, the kept count compared
with the id path's answers -- and, beside it, what a generic decode + re-encode of the same body costs this host (python's json: C code, one thread --
the reference does that with encoding/json before and after its one CheckBulkPermissions).  ACL_DEBUG_LIST=1 prints the call's phase times on stderr."""
import argparse
import ctypes as C
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
ap.add_argument("--calls", type=int, default=20)
ap.add_argument("--sizes", default="1000,10000,65536")
ap.add_argument("--workload", default="c3", choices=["c3", "c4"])
a = ap.parse_args()
w = workloads.c4() if a.workload == "c4" else workloads.c3(batch=65536)
eng = aclgpu.Engine(w.schema, contexts=3, eager_contexts=True)
bench.name_objects(eng, w)
w.load(eng)
eng.snapshot()
rt, perm_name, st = w.check
names = w.names


pod_json = bench.pod_json


template = f"{rt}:{{{{namespacedName}}}}#{perm_name}@{st}:{{{{user.name}}}}"
grants = sorted({int(x) for x in w.res[:65536:2731]})[:24]
eng.write([(aclgpu.OP_TOUCH, (rt, names[rt][g], "viewer", st, "user-sparse", "")) for g in grants])
for m in [int(x) for x in a.sizes.split(",")]:
    m = min(m, len(w.res))
    res = w.res[:m]
    body = json.dumps({"apiVersion": "v1", "kind": "PodList", "metadata": {"resourceVersion": "123456"}, "items": [pod_json(names[rt][int(r)], k) for k, r in enumerate(res)]},
                      separators=(",", ":")).encode()
    t1 = time.perf_counter()
    doc = json.loads(body)
    t_dec = time.perf_counter() - t1
    t1 = time.perf_counter()
    json.dumps(doc, separators=(",", ":"))
    t_enc = time.perf_counter() - t1
    del doc
    dense = int(w.lookup_subjects[0]) if a.workload == "c3" else int(w.subj[0])
    for who, u in (("dense_user", dense), ("sparse_user", None)):
        uname = "user-sparse" if u is None else names[st][u]
        if u is None:
            want = np.isin(res, np.asarray(grants, dtype=res.dtype))
        else:
            tp, te = eng.check_bulk_ids(eng.make_items(rt, perm_name, res, st, "", np.full(m, u, dtype=np.uint32)))
            want = (tp == 2) & (te == 0)
        row = {"items": m, "body_MB": round(len(body) / 1e6, 2), "kept": int(want.sum()), "generic_decode_ms": round(1e3 * t_dec, 1), "generic_encode_ms": round(1e3 * t_enc, 1)}
        out, kept, total = eng.filter_list_response(body, [template], uname)
        kept_names = [it["metadata"]["namespace"] + "/" + it["metadata"]["name"] for it in (json.loads(out)["items"] or [])]
        ok = kept == int(want.sum()) and total == m and kept_names == [names[rt][int(r)] for r, k_ in zip(res, want) if k_]
        ts = []
        arr = (C.c_char_p * 1)(template.encode())
        outp, outn, k_, t_ = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
        for _ in range(a.calls):  # (the C call itself: the mirror's copy of the output into a Python bytes object is the harness's, not the engine's)
            t1 = time.perf_counter()
            rc = eng._L.acl_filter_list_response(eng._h, body, len(body), arr, 1, uname.encode(), C.byref(outp), C.byref(outn), C.byref(k_), C.byref(t_))
            ts.append(time.perf_counter() - t1)
            assert rc == 0 and k_.value == kept
            eng._L.acl_free(outp)
        row["postfilter"] = {"ms": round(1e3 * float(np.median(ts)), 3), "best_ms": round(1e3 * min(ts), 3), "M_items_per_s": round(m / float(np.median(ts)) / 1e6, 2),
                             "body_MB_per_s": round(len(body) / float(np.median(ts)) / 1e6, 1), "equal_to_id_path": bool(ok)}
        # the PreFilter form: one LookupResources for the user (the reference runs it beside the upstream request), then the body against the bitmap
        if True:  # (both kinds of user)
            sid = eng.intern(st, uname) if u is None else u
            t1 = time.perf_counter()
            bms, _cnt = eng.lookup_ids_batch(rt, perm_name, st, "", [sid])
            bm = bms[0]
            t_lookup = time.perf_counter() - t1
            out2, kept2, total2 = eng.prefilter_response(rt, bm, "{{namespacedName}}", eng.BODY_LIST, body)
            ok2 = kept2 == int(want.sum()) and total2 == m and out2 == out
            ts = []
            bmc = np.ascontiguousarray(bm, dtype=np.uint32)
            for _ in range(a.calls):
                t1 = time.perf_counter()
                rc = eng._L.acl_prefilter_response(eng._h, eng.type_id(rt), bmc.ctypes.data, bmc.size, b"{{namespacedName}}", eng.BODY_LIST, body, len(body), C.byref(outp), C.byref(outn),
                                                   C.byref(k_), C.byref(t_))
                ts.append(time.perf_counter() - t1)
                assert rc == 0 and k_.value == kept2
                eng._L.acl_free(outp)
            row["prefilter"] = {"lookup_ms": round(1e3 * t_lookup, 3), "ms": round(1e3 * float(np.median(ts)), 3), "M_items_per_s": round(m / float(np.median(ts)) / 1e6, 2),
                                "body_MB_per_s": round(len(body) / float(np.median(ts)) / 1e6, 1), "equal_to_postfilter_body": bool(ok2)}
        print(m, who, json.dumps(row), flush=True)
