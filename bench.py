#!/usr/bin/env python3
"""bench.py -- Check decisions/s of the MI355X ACL engine on BASELINE.json's headline workload.

A "step" = one pass of the hot path (bulk Check through the C ABI, acl_check_bulk_ids_device)
over one HBM-resident batch of interned requests.  Default workload: C4 (10 M relationships /
1 M objects, 5-level nested groups, 256 k-item batch) -- the configuration BASELINE.json's
metric is quoted on.  N > 1: one process per GPU (torchrun), every rank holds a full replica of
the graph and answers its own batch (weak scaling, no data-path collective: requests are
independent -- SURVEY.md 8(e)); the timed region is bracketed by barrier + synchronize and the
MAX over ranks is reported.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (algorithmic bytes / HIP-event kernel
time vs HBM peak) and `cpu_baseline` (the CPU oracle, a single-thread restatement of SpiceDB's
dispatch -- NOT the embedded SpiceDB, which cannot be built here -- timed on a bounded sample of
the same batch on this box's host cores, and used at the same time to verify the GPU answers).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOAD_DESC = {
    "C1": "C1: 1k-object / 10k-relationship flat namespace#view@user graph, Check",
    "C2": "C2: 100k-object / 1M-relationship 3-level (cluster->namespace->pod) graph, 64k-batch Check",
    "C3": "C3: C2 graph + 64 power users, Filter/LookupResources(pod, view, user) returning ~10k allowed IDs per user",
    "C4": "C4: 10M-relationship / 1M-object 5-level nested-group graph, 256k-batch Check",
    "C5": "C5: 100M-relationship graph sharded by object type, mixed stream (90% 256k-batch Check / 10% Filter) with per-level frontier all-gather",
}


def usable_cores():
    """hardware threads this process may run on: affinity mask, capped by the cgroup CPU quota if there is one"""
    c = max(1, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            c = max(1, min(c, int(int(q) / int(per) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return c


def filter_bench(args, w, eng, world, rank):
    """BASELINE config 3 (not the headline): one step = LookupResources(pod, view, user:U) for the 64 power users as ONE
    batched reverse walk (acl_lookup_resources_batch: bitmaps come back to the host, as the Go side consumes them)."""
    import torch
    rt, perm_name, st = w.check
    subs = np.asarray(w.lookup_subjects, dtype=np.uint32)
    for _ in range(args.warmup):
        eng.lookup_ids_batch(rt, perm_name, st, "", subs)
    eng.stats_reset()
    eng.set_timing(True)
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        bms, counts = eng.lookup_ids_batch(rt, perm_name, st, "", subs)
        lat.append(time.perf_counter() - t1)
    el = time.perf_counter() - t0
    eng.set_timing(False)
    stats = eng.stats()
    # single-request latency (the proxy's shape: one prefilter per list request)
    one = []
    for s_ in subs[:16]:
        t1 = time.perf_counter()
        eng.lookup_ids_batch(rt, perm_name, st, "", [int(s_)])
        one.append(time.perf_counter() - t1)
    out = {"metric": "lookup_resources_per_sec", "value": subs.size * args.steps / el, "unit": "lookups/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic",
           "config": {"workload": WORKLOAD_DESC["C3"], "lookups_per_step": int(subs.size), "relationships": w.ntuples,
                      "objects": int(sum(w.nobjects.values())), "scale": args.scale},
           "allowed_ids_per_lookup": float(np.mean(counts)), "allowed_ids_per_sec": float(np.sum(counts)) * args.steps / el,
           "p50_batch_ms": 1e3 * float(np.median(lat)), "p50_single_lookup_ms": 1e3 * float(np.median(one)),
           "kernel_ms_per_step": stats["kernel_ms"] / args.steps, "rev_expand_launches_per_step": stats["expand_launches"] / args.steps,
           "bitmap_bytes_per_lookup": int(bms.shape[1] * 4)}
    if not args.no_cpu:
        from oracle import orc
        o = orc.Oracle(w.schema)
        w.load(o)
        o.freeze()
        t1 = time.perf_counter()
        want = [np.sort(o.lookup_ids(rt, perm_name, st, "", int(s_))) for s_ in subs[:2]]
        t_cpu = time.perf_counter() - t1
        mism = sum(int(not np.array_equal(np.flatnonzero(np.unpackbits(bms[i].view(np.uint8), bitorder="little")).astype(np.uint32), want[i])) for i in range(2))
        out["parity"] = {"lookups_checked_against_oracle": 2, "mismatches": mism}
        out["cpu_baseline"] = {"value": 2 / t_cpu, "unit": "lookups/s", "cores": 1, "kind": "port",
                               "sample": "2 power users, brute-force definition {id : Check == HAS} over every pod, restated CPU oracle", "seconds": round(t_cpu, 2)}
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if out.get("parity", {}).get("mismatches"):
        raise SystemExit("PARITY FAILURE: GPU lookup differs from the oracle")


def c5_bench(args, w, world, rank, local_rank, t_gen):
    """BASELINE config 5 (not the headline): the graph sharded by object type over G shards, mixed Check + Filter stream.
    G = the ranks of a multi-GPU launch (RCCL all-gather), or --logical-shards (default 8) on one GPU -- the latter is an
    EMULATION of the layout on one device, labelled as such."""
    import torch
    import torch.distributed as dist

    import aclgpu
    from aclgpu import sharded, workloads

    rt, perm_name, st = w.check
    n = int(w.res.size)
    G = world if world > 1 else (args.logical_shards or 8)
    ops = workloads.c5_stream(w, args.steps)
    nC = sum(1 for o in ops if o == "C")
    c5_exchange = "alltoall" if args.exchange == "alltoall" else "allgather"  # Check frontiers; LookupResources always all-gathers
    engines = []

    def make(r, g):
        e = aclgpu.Engine(w.schema, device=local_rank)
        w.load(e)
        engines.append(e)
        return sharded.GpuShard(e, r, g)

    def run(se, barrier):
        e = se.shard.e
        se._alloc(1 << 20)
        items = e.make_items(rt, perm_name, w.res, st, "", w.subj)
        d_items = torch.from_numpy(items.view(np.uint8).copy()).to(se.shard.device)
        p, er = se.check_bulk_ids(d_items)  # warm-up (also builds + uploads the shard's snapshot)
        bm = se.lookup_ids_batch(rt, perm_name, st, "", [int(w.lookup_subjects[0])])
        barrier()
        t0 = time.perf_counter()
        tc = tf = 0.0
        for o in ops:
            t1 = time.perf_counter()
            if o == "C":
                p, er = se.check_bulk_ids(d_items)
                tc += time.perf_counter() - t1
            else:
                bm = se.lookup_ids_batch(rt, perm_name, st, "", [o[1]])
                tf += time.perf_counter() - t1
        barrier()
        el = time.perf_counter() - t0
        st_ = e.stats()
        # cross-path property at full size: the last Filter bitmap must agree with sharded Checks of the same subject
        last_f = [o for o in ops if o != "C"][-1][1]
        rng = np.random.default_rng(7)
        pods = rng.integers(0, w.nobjects[rt], size=20000).astype(np.uint32)
        bm = se.lookup_ids_batch(rt, perm_name, st, "", [last_f])
        bits = np.unpackbits(bm[0].cpu().numpy().view(np.uint8), bitorder="little")
        cp, _ = se.check_bulk_ids(e.make_items(rt, perm_name, pods, st, "", np.full(pods.size, last_f, dtype=np.uint32)))
        cross = int(((cp.cpu().numpy() == 2) != (bits[pods] == 1)).sum())
        return {"elapsed": el, "check_s": tc, "filter_s": tf, "perm": p.cpu().numpy(), "err": er.cpu().numpy(), "cross_mismatch": cross,
                "allowed_last_filter": int(bits.sum()), "local_relationships": int(st_["snapshot_edges_local"]), "recv": se.exchanged_entries,
                "levels": se.levels_last}

    t0 = time.time()
    if world > 1:
        se = sharded.ShardedEngine(make(rank, world), sharded.TorchComm(device=f"cuda:{local_rank}"), exchange=c5_exchange)
        o = run(se, dist.barrier)
        outs = [None] * world
        dist.all_gather_object(outs, {k: v for k, v in o.items() if k not in ("perm", "err")})
        outs[rank].update(perm=o["perm"], err=o["err"])
    else:
        def fn(se):
            return run(se, se.comm.barrier)
        fn.exchange = c5_exchange
        outs = sharded.run_logical_shards(G, make, fn)
    t_all = time.time() - t0
    if rank == 0:
        el = max(o["elapsed"] for o in outs)
        out = {"metric": "mixed_stream_check_decisions_per_sec", "value": n * nC / el, "unit": "decisions/s", "n_gpus": world, "steps": args.steps,
               "warmup": 1, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u32", "data": "synthetic",
               "config": {"workload": WORKLOAD_DESC["C5"], "shards": G, "scale": args.scale, "relationships": w.ntuples,
                          "objects": int(sum(w.nobjects.values())), "check_batch": n, "stream": "".join("C" if o == "C" else "F" for o in ops), "check_exchange": c5_exchange,
                          "execution": "one shard per GPU, RCCL all-gather" if world > 1 else f"{G} LOGICAL shards on ONE GPU: emulated, not a multi-GPU measurement"},
               "check_batches": nC, "filter_requests": len(ops) - nC,
               "ms_per_check_batch": 1e3 * outs[0]["check_s"] / max(1, nC), "ms_per_filter_request": 1e3 * outs[0]["filter_s"] / max(1, len(ops) - nC),
               "levels": outs[0]["levels"], "shard_relationships": [o["local_relationships"] for o in outs],
               "recv_entries_by_shard": [o["recv"] for o in outs], "allowed_ids_last_filter": outs[0]["allowed_last_filter"],
               "setup_s": {"generate": round(t_gen, 1), "load+run": round(t_all, 1)}}
        parity = {"filter_vs_check_cross_mismatches": int(sum(o["cross_mismatch"] for o in outs)), "cross_checked_pods": 20000}
        if not args.no_cpu:
            from oracle import orc
            o = orc.Oracle(w.schema)
            w.load(o)
            o.freeze()
            m = min(n, 4096)
            op_, oe_ = o.check_bulk_ids_mt(usable_cores() if usable_cores() <= 16 else 16, rt, perm_name, w.res[:m], st, "", w.subj[:m])
            parity["checked_against_oracle"] = m
            parity["mismatches"] = int((op_ != outs[0]["perm"][:m]).sum() + (oe_ != outs[0]["err"][:m]).sum())
        out["parity"] = parity
        print(json.dumps(out))
    for e in engines:
        e.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and (out["parity"].get("mismatches") or out["parity"]["filter_vs_check_cross_mismatches"]):
        raise SystemExit("PARITY FAILURE in the sharded mixed stream")


def sharded_leg(args, w, replica, res, subj, world, rank, local_rank, result):
    """The north star's multi-GPU layout: rows partitioned by fnv1a(object type) mod G, one shard per
    rank, per-level all-gather of cross-shard frontier entries over RCCL.  All ranks answer ONE batch together; the
    answers are compared with the replica engine's.  world == 1: G logical shards (threads) on this device -- emulated."""
    import threading

    import torch
    import torch.distributed as dist

    import aclgpu
    from aclgpu import sharded

    rt, perm_name, st = w.check
    n = int(res.size)
    steps = max(2, min(args.steps, 10))
    items = replica.make_items(rt, perm_name, res, st, "", subj)
    want_p, want_e = replica.check_bulk_ids(items)
    G = world if world > 1 else args.logical_shards

    modes = ["allgather", "alltoall"] if args.exchange == "both" else [args.exchange]

    def run(se, comm_barrier, after_mode=None):
        d_items = torch.from_numpy(items.view(np.uint8).copy()).to(se.shard.device)
        res_by_mode = {}
        for mode in modes:
            se.exchange = mode
            se._alloc(max(se.cap, 1 << 20))
            p = e = None
            for _ in range(2):
                p, e = se.check_bulk_ids(d_items)
            x0, c0 = se.exchanged_entries, se.exchanges
            comm_barrier()
            t0 = time.perf_counter()
            lat = []
            for _ in range(steps):
                t1 = time.perf_counter()
                p, e = se.check_bulk_ids(d_items)
                lat.append(time.perf_counter() - t1)
            comm_barrier()
            el = time.perf_counter() - t0
            mism = int((p.cpu().numpy() != want_p).sum() + (e.cpu().numpy() != want_e).sum())
            res_by_mode[mode] = {"elapsed": el, "lat": lat, "mismatches": mism, "levels": se.levels_last,
                                 "recv_entries_per_batch": (se.exchanged_entries - x0) / steps, "exchanges_per_batch": (se.exchanges - c0) / steps}
            if after_mode:
                after_mode(mode, res_by_mode[mode])
        return {"modes": res_by_mode, "shard_relationships": None}

    result.update({
        "layout": f"fnv1a(object type) mod {G}; 16 B frontier entries cross shards once per level",
        "transport": "RCCL over xGMI (torch.distributed nccl)" if world > 1 else "in-process copies between logical shards on ONE GPU (emulated, not a multi-GPU measurement)",
        "shards": G, "batch": n, "steps": steps})

    def mode_done(mode, ms):  # ms: every rank's record of one exchange form (filled in as soon as that form has run)
        el = max(m["elapsed"] for m in ms)
        if True:
            result[mode] = {"collective": "all_gather_into_tensor of every export buffer, each rank keeps what it owns" if mode == "allgather"
                            else "exports grouped by owner on the device, batch_isend_irecv (grouped send/recv) to the owners only",
                            "decisions_per_s": n * steps / el, "ms_per_batch": 1e3 * el / steps, "p50_batch_ms": 1e3 * float(np.median(ms[0]["lat"])),
                            "levels": ms[0]["levels"], "exchanges_per_batch": ms[0]["exchanges_per_batch"],
                            "recv_entries_per_batch_by_shard": [m["recv_entries_per_batch"] for m in ms],
                            "mismatches_vs_replica": int(sum(m["mismatches"] for m in ms))}
        result["decisions_per_s"] = result[modes[0]]["decisions_per_s"]
        result["mismatches_vs_replica"] = int(sum(result[m]["mismatches_vs_replica"] for m in modes if m in result))

    def done(outs):
        result["shard_relationships"] = [o["shard_relationships"] for o in outs]
        for mode in modes:
            mode_done(mode, [o["modes"][mode] for o in outs])

    if world > 1:
        e2 = aclgpu.Engine(w.schema, device=local_rank)
        w.load(e2)
        sh = sharded.GpuShard(e2, rank, world)
        se = sharded.ShardedEngine(sh, sharded.TorchComm(device=f"cuda:{local_rank}"), export_entries=1 << 20)
        def after_mode(mode, rec):  # the all-gather form's numbers survive whatever the all-to-all form does on real RCCL
            g = [None] * world
            dist.all_gather_object(g, rec)
            if rank == 0:
                mode_done(mode, g)

        try:
            o = run(se, dist.barrier, after_mode)
            gathered = [None] * world
            dist.all_gather_object(gathered, int(_local_edges(e2)))
            result["shard_relationships"] = gathered
        finally:
            e2.close()
    else:
        engines = []

        def make(r, g):
            e2 = aclgpu.Engine(w.schema, device=local_rank)
            w.load(e2)
            engines.append((r, e2))
            return sharded.GpuShard(e2, r, g)

        def fn(se):
            o = run(se, se.comm.barrier)
            o["shard_relationships"] = int(_local_edges(se.shard.e))
            return o

        try:
            done(sharded.run_logical_shards(G, make, fn))
        finally:
            for _r, e2 in engines:
                e2.close()
    return result


def _local_edges(engine):
    return engine.stats().get("snapshot_edges_local", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C4", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU-oracle sample time (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sharded", default="auto", choices=["auto", "on", "off"],
                    help="extra leg: the SAME graph partitioned by type hash over the ranks, per-level RCCL all-gather of cross-shard "
                         "frontiers (north star's 8-GPU layout).  auto = on at 8 ranks.  Reported beside `value`, never as `value`.")
    ap.add_argument("--logical-shards", type=int, default=0, help="with 1 GPU: run the sharded leg as G logical shards on this device (emulated)")
    ap.add_argument("--exchange", default="both", choices=["allgather", "alltoall", "both"],
                    help="sharded leg: how Check frontiers cross shards (allgather = the north star's form; alltoall moves G x fewer bytes)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU evaluation path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import aclgpu
    from aclgpu import workloads

    kw = {}
    if args.workload in ("C2", "C3", "C4", "C5"):
        kw["scale"] = args.scale
        if args.batch:
            kw["batch"] = args.batch
    t0 = time.time()
    w = workloads.by_name(args.workload, **kw)
    canon_res, canon_subj = w.res.copy(), w.subj.copy()  # the sharded leg answers ONE stream with all ranks together
    if world > 1:  # every rank answers a different request stream against the same graph
        rng = np.random.default_rng(0x5ACE0000 + rank)
        perm = rng.permutation(w.res.size)
        w.res = w.res[perm]
        w.subj = np.roll(w.subj[perm], rank * 7919) if rank else w.subj[perm]
    t_gen = time.time() - t0
    rt, perm_name, st = w.check
    n = int(w.res.size)

    if args.workload == "C5":
        return c5_bench(args, w, world, rank, local_rank, t_gen)
    eng = aclgpu.Engine(w.schema, device=local_rank)
    t0 = time.time()
    w.load(eng)
    eng.snapshot()
    t_load = time.time() - t0
    if args.workload == "C3":
        return filter_bench(args, w, eng, world, rank)
    items = eng.make_items(rt, perm_name, w.res, st, "", w.subj)
    d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
    d_perm = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_err = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def step():
        eng.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
        eng.sync()

    for _ in range(args.warmup):
        step()
    eng.stats_reset()
    eng.set_timing(True)  # HIP events around every engine kernel, on the engine's own stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lat = []
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        step()
        lat.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_begin
    eng.set_timing(False)
    stats = eng.stats()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    gpu_perm = d_perm.cpu().numpy()
    gpu_err = d_err.cpu().numpy()
    # (ii) of SURVEY.md 8(d): the ABI call that takes HOST buffers -- H2D of the items, kernels, D2H of perm/err (never `value`)
    host_lat = []
    for _ in range(5):
        t1 = time.perf_counter()
        eng.check_bulk_ids(items)
        host_lat.append(time.perf_counter() - t1)

    # ---- extra leg (outside the timed region above): the sharded graph
    want_sharded = (args.sharded == "on" or (args.sharded == "auto" and world == 8)) and (world > 1 or args.logical_shards > 1)

    out = None
    if rank == 0:
        total = n * args.steps * world
        launches = max(1, stats["expand_launches"])
        out = {
            "metric": "check_decisions_per_sec", "value": total / elapsed, "unit": "decisions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC[args.workload], "batch_per_gpu": n, "relationships": w.ntuples,
                       "objects": int(sum(w.nobjects.values())), "scale": args.scale, "parallelism": f"replicas x{world} (request-level data parallel)"},
            "p50_batch_ms": 1e3 * float(np.median(lat)), "p95_batch_ms": 1e3 * float(np.percentile(lat, 95)),
            "has_fraction": float((gpu_perm == 2).mean()), "levels": int(stats["levels_last"]),
            "expand_launches_per_batch": launches / args.steps, "kernel_ms_per_batch": stats["kernel_ms"] / args.steps,
            "setup_s": {"generate": round(t_gen, 2), "load+snapshot": round(t_load, 2)},
            "snapshot_bytes": int(stats["snapshot_bytes"]),
            "host_buffer_path": {"p50_batch_ms": 1e3 * float(np.median(host_lat)), "decisions_per_s": n / float(np.median(host_lat)),
                                 "note": "acl_check_bulk_ids: pageable host items in, perm+err out (PCIe inclusive)"},
        }
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "k_expand", "kernel_avg_us": 1e3 * stats["expand_ms"] / launches}
        cpu = None
        if world == 1 and not args.no_cpu:
            from oracle import orc
            t0 = time.time()
            o = orc.Oracle(w.schema)
            w.load(o)
            o.freeze()
            t_oload = time.time() - t0
            # calibrate, then time a bounded sample of the SAME batch (prefix), single thread
            m0 = min(n, 512)
            t0 = time.perf_counter()
            o.check_bulk_ids(rt, perm_name, w.res[:m0], st, "", w.subj[:m0])
            per = (time.perf_counter() - t0) / m0
            m = int(min(n, max(m0, args.cpu_seconds / max(per, 1e-9))))
            t0 = time.perf_counter()
            operm, oerr = o.check_bulk_ids(rt, perm_name, w.res[:m], st, "", w.subj[:m])
            t_cpu = time.perf_counter() - t0
            mism = int((operm != gpu_perm[:m]).sum() + (oerr != gpu_err[:m]).sum())
            out["parity"] = {"checked_against_oracle": m, "mismatches": mism}
            # all host cores: the same oracle, the whole batch split statically over threads (SURVEY.md 8(d) "CPU baseline beside it" (b))
            cores = usable_cores()
            # the box may expose more hardware threads than this container may use: keep the thread count that is fastest
            best = (0.0, 1)
            cm = min(n, 16384)
            c_try = 4
            while c_try <= cores:
                t0 = time.perf_counter()
                o.check_bulk_ids_mt(c_try, rt, perm_name, w.res[:cm], st, "", w.subj[:cm])
                r_ = cm / (time.perf_counter() - t0)
                if r_ > best[0]:
                    best = (r_, c_try)
                c_try *= 2
            cores = best[1]
            mm = int(min(n, max(m, best[0] * args.cpu_seconds)))
            t0 = time.perf_counter()
            mperm, merr = o.check_bulk_ids_mt(cores, rt, perm_name, w.res[:mm], st, "", w.subj[:mm])
            t_mt = time.perf_counter() - t0
            mism_mt = int((mperm != gpu_perm[:mm]).sum() + (merr != gpu_err[:mm]).sum())
            out["parity"]["checked_against_oracle"] = max(m, mm)
            out["parity"]["mismatches"] = mism + mism_mt
            cpu = {"value": mm / t_mt, "unit": "decisions/s", "cores": cores, "kind": "port",
                   "sample": f"first {mm} of the {n}-item batch split statically over {cores} host threads, restated CPU oracle (not embedded SpiceDB)",
                   "seconds": round(t_mt, 2), "load_s": round(t_oload, 2),
                   "single_thread": {"value": m / t_cpu, "sample": f"first {m} items", "seconds": round(t_cpu, 2)}}
            # algorithmic bytes per Check (SURVEY.md 8(d) model) from the oracle's counter on a sub-sample
            mb = min(m, 4096)
            tot = 0
            for i in range(mb):
                b, _r = o.check_bytes(rt, perm_name, int(w.res[i]), st, "", int(w.subj[i]))
                tot += b
            bytes_per_check = tot / mb
            batch_bytes = bytes_per_check * n
            ach = batch_bytes * args.steps / (stats["expand_ms"] * 1e-3) / 1e9 if stats["expand_ms"] > 0 else None
            roof.update({"achieved": ach, "frac": (ach / HBM_PEAK_GBS) if ach else None, "algorithmic_bytes_per_check": bytes_per_check,
                         "algorithmic_bytes_per_launch": batch_bytes * args.steps / launches})
            tr = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tr):
                try:
                    roof["traffic"] = json.load(open(tr)).get(args.workload)
                except Exception:  # noqa: BLE001
                    pass
        out["roofline"] = roof
        out["cpu_baseline"] = cpu
    # ---- extra leg (outside the timed region, after the main line is complete): the sharded graph.  Whatever happens in
    # it -- an exception on this rank, a wedged collective -- the main line is still printed exactly once.
    if want_sharded:
        printed = threading.Event()

        def emit(extra):
            if rank == 0 and not printed.is_set():
                printed.set()
                out["sharded"] = extra
                print(json.dumps(out), flush=True)

        sharded_result = {}

        def on_timeout():
            sharded_result["error"] = "sharded leg did not finish within 180 s (collective wedged?)"
            emit(sharded_result)
            os._exit(0)

        timer = threading.Timer(180.0, on_timeout)
        timer.daemon = True
        timer.start()
        try:
            emit(sharded_leg(args, w, eng, canon_res, canon_subj, world, rank, local_rank, sharded_result))
        except Exception as ex:  # noqa: BLE001
            sharded_result["error"] = f"{type(ex).__name__}: {ex}"
            emit(sharded_result)
        finally:
            timer.cancel()
    elif rank == 0:
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out and out.get("parity", {}).get("mismatches"):
        raise SystemExit("PARITY FAILURE: GPU answers differ from the oracle")


if __name__ == "__main__":
    main()
