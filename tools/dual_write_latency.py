#!/usr/bin/env python3
"""The reference's REAL dual-write sequence against the full-size C4 graph (10 M relationships), across a background compaction.

Per kube write the proxy's pessimistic workflow issues two WriteRelationships (pkg/authz/distributedtx):
  W1  preconditions [MUST_NOT_MATCH lock:<hash>#workflow@workflow:*]               (workflow.go:452-462)
      updates       payload (CREATE pod#creator, TOUCH pod#namespace) + CREATE lock:<hash>#workflow@workflow:<id>   (workflow.go:134-176, 392-418)
                    + CREATE workflow:<id>#idempotency_key@activity:<payload hash>, expiring                        (activity.go:54-102)
  ... the kube write ...
  W2  updates       DELETE lock:<hash>#workflow@workflow:<id> + CREATE another expiring idempotency key            (workflow.go:86-129, 238)
and every request that follows reads fully consistent (check.go:41-46).  On a quiet proxy the `lock#workflow` class is EMPTY whenever a snapshot
is (re)built, so W1 lands in an empty class every time: that used to force a synchronous rebuild of the whole snapshot (100-140 ms here).
Timed: the Check right after W1 (the creator gets its pod), the Check right after W2, a concurrent writer's conflict; the clock advances
so that idempotency keys keep expiring under the reads.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
w = workloads.c4(scale=scale)
schema = "use expiration\n" + w.schema + """
definition lock {
  relation workflow: workflow
}
definition workflow {
  relation idempotency_key: activity with expiration
}
definition activity {}
"""
e = aclgpu.Engine(schema)
w.load(e)
now = 1_000_000
e.set_now(now)
e.snapshot()
e.lookup("pod", "view", "user", "nobody")  # reverse rows built + uploaded once
st0 = e.stats()
KEY_TTL = 150  # seconds; the clock advances 1 s per kube write, so a key runs out under the reads every iteration once the first 150 are through


def kube_create(i, tag, lat1, lat2, wl):
    global now
    pod, user, wf = f"dw/{tag}{i}", f"paul{i % 500}", f"wf-{tag}{i}"
    lock = ("lock", f"{hash((tag, i)) & 0xFFFFFFFFFFFFFFFF:x}", "workflow", "workflow", wf, "")
    pre = [(aclgpu.PRE_MUST_NOT_MATCH, dict(rtype="lock", rid=lock[1], rel="workflow", stype="workflow"))]
    t0 = time.perf_counter()
    e.write([(aclgpu.OP_CREATE, ("pod", pod, "creator", "user", user, "")), (aclgpu.OP_TOUCH, ("pod", pod, "namespace", "namespace", "ns", "")),
             (aclgpu.OP_CREATE, lock), (aclgpu.OP_CREATE, ("workflow", wf, "idempotency_key", "activity", f"a1-{tag}{i}", ""), now + KEY_TTL)], pre)
    t1 = time.perf_counter()
    ok = e.check("pod", pod, "view", "user", user) == (2, 0)
    t2 = time.perf_counter()
    conflict = None
    if i % 50 == 0:  # a second writer of the same object while the lock is held: must fail, and change nothing
        try:
            e.write([(aclgpu.OP_TOUCH, ("pod", pod, "creator", "user", "mallory", "")), (aclgpu.OP_CREATE, ("lock", lock[1], "workflow", "workflow", wf + "-2", ""))], pre)
            conflict = False
        except aclgpu.AclError as ex:
            conflict = ex.code == aclgpu.ERR_FAILED_PRECONDITION
        ok = ok and e.check("pod", pod, "view", "user", "mallory") == (1, 0)
    t3 = time.perf_counter()
    e.write([(aclgpu.OP_DELETE, lock), (aclgpu.OP_CREATE, ("workflow", wf, "idempotency_key", "activity", f"a2-{tag}{i}", ""), now + KEY_TTL)])
    t4 = time.perf_counter()
    ok = ok and e.check("pod", pod, "view", "user", user) == (2, 0)
    t5 = time.perf_counter()
    lat1.append(t2 - t1)
    lat2.append(t5 - t4)
    wl.append((t1 - t0) + (t4 - t3))
    now += 1
    e.set_now(now)
    return ok, conflict


# ---- phase 1: 400 kube writes on the freshly built snapshot (the lock class is empty in it)
l1, l2, wl = [], [], []
oks, confl = [], []
for i in range(400):
    ok, c = kube_create(i, "a", l1, l2, wl)
    oks.append(ok)
    if c is not None:
        confl.append(c)
st1 = e.stats()
# ---- phase 2 (untimed): bulk creates until the pod tables' headroom is 88 % used, every patch kept small by a read after each write
nrows = int(w.nobjects["pod"] * 1.25) + 1024  # plan.cpp with_headroom
target = int(nrows * 0.88)
k = 0
while e.object_count("pod") < target:
    m = min(500, target - e.object_count("pod"))
    ups = []
    for _ in range(m):
        ups.append((aclgpu.OP_TOUCH, ("pod", f"bulk/{k}", "creator", "user", f"paul{k % 500}", "")))
        ups.append((aclgpu.OP_TOUCH, ("pod", f"bulk/{k}", "namespace", "namespace", "ns", "")))
        k += 1
    e.write(ups)
    e.check("pod", "bulk/0", "view", "user", "paul0")
st2 = e.stats()
# ---- phase 3 (timed): kube writes across the 90 % mark that starts a background build and on past its adoption -- in the adopted snapshot the lock
# class is empty again (every lock was deleted by its W2), so the first W1 after the adoption is the case that used to rebuild
c1, c2, cw = [], [], []
nb = int(nrows * 0.035)
for i in range(nb):
    ok, c = kube_create(i, "b", c1, c2, cw)
    oks.append(ok)
    if c is not None:
        confl.append(c)
st3 = e.stats()
objects_end = {t: e.object_count(t) for t in ("lock", "workflow", "activity", "pod")}
lk0 = time.perf_counter()
got = e.lookup("pod", "view", "user", "paul7")
lk = time.perf_counter() - lk0
e.close()


def pct(a):
    a = 1e3 * np.asarray(a)
    return {"p50_ms": round(float(np.median(a)), 4), "p99_ms": round(float(np.percentile(a, 99)), 4), "p999_ms": round(float(np.percentile(a, 99.9)), 4), "max_ms": round(float(a.max()), 3),
            "over_1ms": int((a > 1.0).sum())}


print(json.dumps({
    "workload": f"C4 x{scale}: {w.ntuples} relationships + lock / workflow / activity definitions (bootstrap.yaml:30-38)",
    "sequence": "W1 [MUST_NOT_MATCH lock] {payload, CREATE lock, CREATE expiring idempotency key} -> Check -> W2 {DELETE lock, CREATE expiring key} -> Check; clock +1 s per kube write, keys live 150 s",
    "all_reads_correct": bool(all(oks)), "lock_conflicts_detected": f"{sum(confl)}/{len(confl)}",
    "fresh_snapshot": {"kube_writes": 400, "check_after_W1": pct(l1), "check_after_W2": pct(l2), "two_writes_ms_p50": round(1e3 * float(np.median(wl)), 4),
                       "synchronous_rebuilds": st1["snapshot_builds"] - st0["snapshot_builds"], "patches": st1["snapshot_patches"] - st0["snapshot_patches"]},
    "across_compaction": {"bulk_creates_before": k, "kube_writes": nb, "check_after_W1": pct(c1), "check_after_W2": pct(c2), "two_writes_ms_p50": round(1e3 * float(np.median(cw)), 4),
                          "snapshot_compactions": st3["snapshot_compactions"] - st2["snapshot_compactions"],
                          "synchronous_rebuilds": st3["snapshot_builds"] - st2["snapshot_builds"], "patches": st3["snapshot_patches"] - st2["snapshot_patches"]},
    # object ids are recycled (round 4): a lock is free again behind its W2, a workflow / its activities once their keys are collected (24 h) --
    # with the production quarantine (30 s) a run this short recycles little; ACL_ID_QUARANTINE_MS shortens it for the run
    "id_recycling": {"quarantine_ms": int(os.environ.get("ACL_ID_QUARANTINE_MS", "30000")), "ids_recycled": int(st3.get("ids_recycled", 0)), "objects_at_end": objects_end,
                     "locks_written": 400 + nb},
    "lookup_after_all_ms": round(1e3 * lk, 3), "lookup_ids": len(got)}))
if not all(oks) or not all(confl):
    raise SystemExit("dual-write run: a read was wrong or a lock conflict went unnoticed")
