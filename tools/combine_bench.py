#!/usr/bin/env python3
"""usage (GPU box): tools/combine_bench.py [scale] -- what a non-monotone permission costs next to a pure-union one on the SAME graph.

The "banned users" schema of tests/test_combine_gpu.py (exclusion over arrows into a non-monotone namespace#view, a non-monotone userset subject
group#active, wildcards on both sides of `-`, an intersection) at `scale` x 30 000 pods; one 262 144-item Check batch per permission, device
resident, HIP-event kernel time: `loose` (pure union: the monotone instantiation of the walk would NOT be used here -- the schema has combine
programs, so every launch is the CMB one) against `view` = loose - banned and `strict` = creator & namespace->view.  A sample of the answers is
checked against the oracle.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import torch  # noqa: E402
import aclgpu  # noqa: E402
from oracle import orc  # noqa: E402
from tests.test_combine_gpu import SCHEMA_BANS, bans_graph, load_numeric  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
E, n = bans_graph(7, n_user=int(4000 * scale), n_group=int(600 * scale), n_ns=int(200 * scale), n_pod=int(30000 * scale))
nrel = sum(len(x[4]) for x in E)
B = 262144
rng = np.random.default_rng(3)
res = rng.integers(0, n["pod"], size=B).astype(np.uint32)
sub = rng.integers(0, n["user"], size=B).astype(np.uint32)
creators = dict(zip(E[8][4].tolist(), E[8][5].tolist()))
for i in range(0, B, 3):
    sub[i] = creators[int(res[i])]
out = {"workload": f"bans graph x{scale}: {nrel} relationships, {n}", "batch": B, "permissions": {}}
with aclgpu.Engine(SCHEMA_BANS) as e:
    load_numeric(e, E)
    co = orc.Oracle(SCHEMA_BANS)
    load_numeric(co, E)
    co.freeze()
    for perm in ("loose", "view", "strict"):
        items = e.make_items("pod", perm, res, "user", "", sub)
        d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
        d_perm = torch.zeros(B, dtype=torch.uint8, device="cuda")
        d_err = torch.zeros(B, dtype=torch.int32, device="cuda")
        for _ in range(3):
            e.check_bulk_ids_device(d_items.data_ptr(), B, d_perm.data_ptr(), d_err.data_ptr())
        torch.cuda.synchronize()
        e.set_timing(True)
        e.stats_reset()
        K = 20
        t0 = time.perf_counter()
        for _ in range(K):
            e.check_bulk_ids_device(d_items.data_ptr(), B, d_perm.data_ptr(), d_err.data_ptr())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = e.stats()
        e.set_timing(False)
        m = 20000
        op, oe = co.check_bulk_ids_mt(16, "pod", perm, res[:m], "user", "", sub[:m])
        got_p, got_e = d_perm.cpu().numpy()[:m], d_err.cpu().numpy()[:m]
        out["permissions"][perm] = {"decisions_per_s": B * K / el, "kernel_us_per_batch": 1e3 * st["kernel_ms"] / K, "single_launch_passes": int(st["local_passes"]),
                                    "level_loop_launches": int(st["expand_launches"]), "has_fraction": float((d_perm.cpu().numpy() == 2).mean()),
                                    "equal_to_oracle_on_sample": bool(np.array_equal(got_p, op) and np.array_equal(got_e, oe))}
    # the monotone instantiation on the same graph: the schema WITHOUT its non-monotone permissions (loose only; group#active replaced by group#member)
    mono = SCHEMA_BANS.replace("permission active = member - banned", "permission active = member").replace("permission view = viewer - banned", "permission view = viewer") \
        .replace("permission view = (viewer + creator + namespace->view) - banned", "permission view = viewer + creator + namespace->view").replace("permission strict = creator & namespace->view", "")
with aclgpu.Engine(mono) as e2:
    load_numeric(e2, E)
    items = e2.make_items("pod", "loose", res, "user", "", sub)
    d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
    d_perm = torch.zeros(B, dtype=torch.uint8, device="cuda")
    d_err = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(3):
        e2.check_bulk_ids_device(d_items.data_ptr(), B, d_perm.data_ptr(), d_err.data_ptr())
    torch.cuda.synchronize()
    e2.set_timing(True)
    e2.stats_reset()
    t0 = time.perf_counter()
    for _ in range(20):
        e2.check_bulk_ids_device(d_items.data_ptr(), B, d_perm.data_ptr(), d_err.data_ptr())
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["monotone_schema_loose"] = {"decisions_per_s": B * 20 / el, "kernel_us_per_batch": 1e3 * e2.stats()["kernel_ms"] / 20, "has_fraction": float((d_perm.cpu().numpy() == 2).mean())}
print(json.dumps(out))
