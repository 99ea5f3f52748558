#!/usr/bin/env python3
"""The reference's dual-write sequence for N kube writes on a STORE-ONLY engine (no GPU): do the id spaces of the scaffolding types stop
growing?  Per kube write: W1 {payload, CREATE lock:<hash>#workflow@workflow:<id>, CREATE an expiring idempotency key}
with the MUST_NOT_MATCH lock precondition, W2 {DELETE lock, CREATE another expiring key} (workflow.go:134-201,392-462, activity.go:54-102);
the clock advances one second per kube write, keys live 150 s and are collected 24 h later (spicedb.go:66).  Every `--check-every` kube
writes the host snapshot is brought up to date the way a read does it (in-place patch, acl_selfcheck_snapshot) and verified against the
store.  Prints one JSON line: object counts over time, ids recycled, RSS, patched-vs-rebuilt snapshot updates.

usage: tools/id_recycling.py [kube_writes=1000000] [--delete-payload]"""
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import aclgpu  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
DELETE_PAYLOAD = "--delete-payload" in sys.argv  # every pod is deleted again 2 000 kube writes later (a cluster in steady state): pod ids recycle too
CHECK_EVERY = 1000
b = json.load(open(os.path.join(ROOT, "tests", "golden", "bootstrap.json")))
e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
now = 1_700_000_000
e.set_now(now)
KEY_TTL = 150
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024  # noqa: E731  (MiB; a high-water mark)
samples, patched, rebuilt = [], 0, 0
t0 = time.time()
for i in range(N):
    pod, user, wf = f"ns{i % 50}/pod-{i}", f"user{i % 500}", f"wf{i}"
    lock = ("lock", f"{(i * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF:x}", "workflow", "workflow", wf, "")
    pre = [(aclgpu.PRE_MUST_NOT_MATCH, dict(rtype="lock", rid=lock[1], rel="workflow", stype="workflow"))]
    e.write([(aclgpu.OP_CREATE, ("pod", pod, "creator", "user", user, "")), (aclgpu.OP_TOUCH, ("pod", pod, "namespace", "namespace", f"ns{i % 50}", "")),
             (aclgpu.OP_CREATE, lock), (aclgpu.OP_CREATE, ("workflow", wf, "idempotency_key", "activity", f"a1x{i}", ""), now + KEY_TTL)], pre)
    e.write([(aclgpu.OP_DELETE, lock), (aclgpu.OP_CREATE, ("workflow", wf, "idempotency_key", "activity", f"a2x{i}", ""), now + KEY_TTL)])
    if DELETE_PAYLOAD and i >= 2000:
        j = i - 2000
        e.delete_by_filter(rtype="pod", rid=f"ns{j % 50}/pod-{j}")
    now += 1
    e.set_now(now)
    if (i + 1) % CHECK_EVERY == 0:
        code = e.selfcheck_snapshot_code()  # raises when the snapshot does not match the store
        patched += code == 1
        rebuilt += code == 0
    if (i + 1) % max(1, N // 20) == 0:
        samples.append({"kube_writes": i + 1, "objects": {t: e.object_count(t) for t in ("lock", "workflow", "activity", "pod")}, "ids_recycled": e.stats()["ids_recycled"],
                        "relationships": len(e.read(rtype="workflow")) + len(e.read(rtype="lock")), "rss_mib": rss(), "elapsed_s": round(time.time() - t0, 1)})
last, mid = samples[-1], samples[len(samples) // 2]
print(json.dumps({"kube_writes": N, "delete_payload": DELETE_PAYLOAD, "seconds": round(time.time() - t0, 1), "snapshot_updates": {"patched_in_place": patched, "rebuilt": rebuilt},
                  "plateau": {t: {"at_half": mid["objects"][t], "at_end": last["objects"][t]} for t in ("lock", "workflow", "activity", "pod")},
                  "rss_mib": {"at_half": mid["rss_mib"], "at_end": last["rss_mib"]}, "ids_recycled": last["ids_recycled"], "samples": samples}))
