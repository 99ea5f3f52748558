#!/usr/bin/env python3
"""bench.py -- Check decisions/s of the MI355X ACL engine on BASELINE.json's headline workload.

A "step" = one pass of the hot path (bulk Check through the C ABI) over one batch of interned requests.  Default
workload: C4 (10 M relationships / 1 M objects, 5-level nested groups, 256 k-item batch) -- the configuration
BASELINE.json's metric is quoted on.  What is timed follows SURVEY.md 8(d):

  value             (ii) the ABI call that takes HOST ids -- items and answers cross PCIe inside the call: K steps over 8 distinct pre-generated
                    batches in pinned host memory, issued by `--callers` (default 3) host threads that each block in
                    acl_check_bulk_ids -- what goroutines behind the cgo shim do.  The kernel reads the items from, and writes the
                    answers to, that pinned memory itself (no copies); the engine admits three chip-filling batches at once and their
                    kernels fill each other's tails (profiles/r03_hostmapped_batches.txt: 1 / 2 / 3 / 4 / 8 / 16 callers
                    692 / 884 / 912 / 893 / 909 / 901 M/s; kernels alone, back to back: 873-889).  `--pipeline submit` times
                    acl_check_bulk_ids_submit / acl_ticket_wait from one thread instead (windows 2-6: 899-945 M/s).
  device_resident   (i) kernels only: the batch is already in HBM (acl_check_bulk_ids_device), sequential; the roofline's
                    per-launch kernel time comes from HIP events in THIS leg (pipelined launches overlap each other)
  latency           p50 / p95 of >= 200 single, unpipelined host-id calls ("batch latency")
  string_path       (iii) acl_check_bulk with strings (5 strings interned per item), reported separately

Sharded leg (8 ranks, or --sharded on): the SAME graph partitioned by type hash, answered by all ranks together; with more than one
rank it runs in child processes with a process group of their own (isolated_sharded_leg), so that nothing an RCCL path does on a real
multi-GPU node can take the headline line with it.

N > 1: one process per GPU (torchrun), every rank holds a full replica of the graph and answers its own batches (weak
scaling, no data-path collective: requests are independent -- SURVEY.md 8(e)); the timed region is bracketed by barrier +
synchronize and the MAX over ranks is reported.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (algorithmic bytes of the WHOLE batch from the oracle's byte model,
multi-threaded, with its per-level split / HIP-event kernel time vs HBM peak; `traffic` = HBM bytes per launch MEASURED IN THE RUN at N=1 --
the device leg of this very command re-run under two rocprofv3 --pmc passes, ~5 s -- or, for the other configurations and N > 1, the
figure of profiles/traffic.json from an earlier run's passes, labelled), `cpu_baseline` (the CPU oracle, a restatement of
SpiceDB's dispatch -- NOT the embedded SpiceDB, which cannot be built here -- timed on a bounded sample of the same batch on
this box's host cores and used at the same time to verify the GPU answers), and `configs` = {C2, C3}: the other single-GPU
BASELINE configurations measured in the same run, each with its own roofline / cpu_baseline / parity; `single_checks` = the proxy's own
call shape (64 / 256 / 1 024 concurrent single CheckPermission calls through the micro-batcher), from tools/bin/batcher_bench.
"""
import argparse
import json
import os
import subprocess
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
    "C5R": "C5-size graph (100M relationships / 10M objects, ~0.7 GB snapshot: beyond the 256 MiB Infinity Cache) as ONE single-GPU replica, 256k-batch Check",
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


def bind_to_gpu_numa_node(device: int):
    """Runs this rank on the NUMA node its GPU hangs off (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node): the kernels read
    the request arrays from, and write the answers to, pinned HOST memory, and the bulk string interning walks host tables -- both are
    allocated first-touch by this process's threads.  On a two-socket node half the ranks would otherwise sit across the socket link from
    their GPU.  Returns the node, or None when anything about it cannot be found out (nothing is changed then)."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        bus = buf.value.decode().strip().lower()
        node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) < 2:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:  # noqa: BLE001
        return None


def c3_lookup_bytes(w, subjects):
    """SURVEY.md 8(d) LookupResources formula, evaluated on the generator's arrays for C3's schema:
    17 + sum over the reverse rows the walk touches of (8 + 4 * deg) + ceil(N_pod / 8) for the result bitmap."""
    E = {(e[0], e[1]): (e[4], e[5]) for e in w.edges}

    def by_subject(key, n):
        r, s_ = E[key]
        o = np.argsort(s_, kind="stable")
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(ptr, s_.astype(np.int64) + 1, 1)
        return np.cumsum(ptr), r[o]

    nu, nns, ncl, npod = w.nobjects["user"], w.nobjects["namespace"], w.nobjects["cluster"], w.nobjects["pod"]
    user_rows = {k: by_subject(k, nu) for k in [("pod", "viewer"), ("pod", "creator"), ("namespace", "viewer"), ("namespace", "creator"), ("cluster", "viewer"), ("cluster", "admin")]}
    ns_of_cl = by_subject(("namespace", "cluster"), ncl)
    pod_of_ns = by_subject(("pod", "namespace"), nns)
    out = []
    for u in subjects:
        u = int(u)
        b = 17 + (npod + 7) // 8
        reach = {}
        for k, (ptr, col) in user_rows.items():
            d = int(ptr[u + 1] - ptr[u])
            b += 8 + 4 * d
            reach.setdefault(k[0], []).append(col[ptr[u]:ptr[u + 1]])
        cls = np.unique(np.concatenate(reach["cluster"])) if reach.get("cluster") else np.zeros(0, dtype=np.int64)
        nss = [np.concatenate(reach["namespace"])] if reach.get("namespace") else []
        for c in cls:
            d = int(ns_of_cl[0][c + 1] - ns_of_cl[0][c])
            b += 8 + 4 * d
            nss.append(ns_of_cl[1][ns_of_cl[0][c]:ns_of_cl[0][c + 1]])
        nss = np.unique(np.concatenate(nss)) if nss else np.zeros(0, dtype=np.int64)
        deg = pod_of_ns[0][nss.astype(np.int64) + 1] - pod_of_ns[0][nss.astype(np.int64)]
        b += int(8 * nss.size + 4 * deg.sum())
        out.append(b)
    return np.asarray(out, dtype=np.int64)


def c3_cpu_reverse_walk(w, subjects):
    """CPU LookupResources for C3's schema: the same reverse walk the device runs (subject -> the rows that name it -> down the
    cluster -> namespace -> pod arrows), over CSR-by-subject arrays in numpy -- the tuned CPU baseline VERDICT r1 asked for next to
    the brute-force definition.  Returns (list of allowed-pod id arrays, seconds for the walks alone: index building is setup)."""
    E = {(e[0], e[1]): (e[4], e[5]) for e in w.edges}

    def by_subject(key, n):
        r, s_ = E[key]
        o = np.argsort(s_, kind="stable")
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(ptr, s_.astype(np.int64) + 1, 1)
        return np.cumsum(ptr), r[o].astype(np.int64)

    nu, nns, ncl, npod = w.nobjects["user"], w.nobjects["namespace"], w.nobjects["cluster"], w.nobjects["pod"]
    rows = {k: by_subject(k, nu) for k in [("pod", "viewer"), ("pod", "creator"), ("namespace", "viewer"), ("namespace", "creator"), ("cluster", "viewer"), ("cluster", "admin")]}
    ns_of_cl = by_subject(("namespace", "cluster"), ncl)
    pod_of_ns = by_subject(("pod", "namespace"), nns)

    def expand(csr, ids):  # all resources of the rows of `ids`
        ptr, col = csr
        if not ids.size:
            return np.zeros(0, dtype=np.int64)
        st_, en = ptr[ids], ptr[ids + 1]
        tot = int((en - st_).sum())
        if not tot:
            return np.zeros(0, dtype=np.int64)
        idx = np.repeat(st_ - np.concatenate(([0], np.cumsum(en - st_)[:-1])), en - st_) + np.arange(tot)
        return col[idx]

    outs = []
    t0 = time.perf_counter()
    for u in subjects:
        one = np.asarray([int(u)], dtype=np.int64)
        cl = np.unique(np.concatenate([expand(rows[("cluster", "viewer")], one), expand(rows[("cluster", "admin")], one)]))
        ns = np.unique(np.concatenate([expand(rows[("namespace", "viewer")], one), expand(rows[("namespace", "creator")], one), expand(ns_of_cl, cl)]))
        seen = np.zeros(npod, dtype=bool)
        for part in (expand(rows[("pod", "viewer")], one), expand(rows[("pod", "creator")], one), expand(pod_of_ns, ns)):
            seen[part] = True
        outs.append(np.flatnonzero(seen))
    return outs, time.perf_counter() - t0


def filter_bench(args, w, eng, steps, warmup):
    """BASELINE config 3: one step = LookupResources(pod, view, user:U) for the 64 power users as ONE batched reverse walk
    (acl_lookup_resources_batch: bitmaps come back to the host, as the Go side consumes them).  Returns the C3 record."""
    import torch
    rt, perm_name, st = w.check
    subs = np.asarray(w.lookup_subjects, dtype=np.uint32)
    # result buffers: allocated once with acl_host_alloc (pinned) and reused across steps, as the Go shim does -- the single-launch walk then
    # writes the result rows straight into them; `pageable` below is the same call with ordinary numpy arrays (staged through the context)
    words = max(1, (eng.object_count(rt) + 31) // 32)
    hb = eng.host_alloc(subs.size * words * 4 + subs.size * 8)
    bufs = (hb[:subs.size * words * 4].view(np.uint32).reshape(subs.size, words), hb[subs.size * words * 4:].view(np.uint64))
    for _ in range(max(1, warmup)):
        eng.lookup_ids_batch(rt, perm_name, st, "", subs, out=bufs)
    eng.stats_reset()
    eng.set_timing(True)
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        bms, counts = eng.lookup_ids_batch(rt, perm_name, st, "", subs, out=bufs)
        lat.append(time.perf_counter() - t1)
    el = time.perf_counter() - t0
    eng.set_timing(False)
    stats = eng.stats()
    assert bms is bufs[0]
    only_batched = args.legs == "device"  # (rocprofv3 runs: every launch of the process is then a 64-lookup launch, so the profiler's average is the roofline's)
    pg_bufs = None
    pg = []
    for _ in range(0 if only_batched else max(5, warmup)):
        t1 = time.perf_counter()
        pg_bufs = eng.lookup_ids_batch(rt, perm_name, st, "", subs, out=pg_bufs)
        pg.append(time.perf_counter() - t1)
    pageable_equal = only_batched or bool(np.array_equal(pg_bufs[0], bms) and np.array_equal(pg_bufs[1], counts))
    if only_batched:
        pg = [float("nan")]
    # single-request latency (the proxy's shape: one prefilter per list request)
    one = []
    one_buf = (bufs[0][:1], bufs[1][:1])
    keep = (bms.copy(), counts.copy())
    for s_ in np.tile(subs, 4)[:0 if only_batched else 200]:
        t1 = time.perf_counter()
        eng.lookup_ids_batch(rt, perm_name, st, "", [int(s_)], out=one_buf)
        one.append(time.perf_counter() - t1)
    if only_batched:
        one = [float("nan")]
    bms, counts = keep
    # ... and "list namespaces" (e2e/proxy_test.go:656-659: LookupResources(namespace, view, user)): a result slot that pod#view expands through its arrow -- the
    # walk is directed at the result slot and ends there (Snapshot::rev_useful / rev_sink), it does not go on into the allowed namespaces' pods
    ns_one = []
    if not only_batched and "namespace" in w.nobjects:
        ns_bufs = None
        for s_ in np.tile(subs, 4)[:100]:
            t1 = time.perf_counter()
            ns_bufs = eng.lookup_ids_batch("namespace", perm_name, st, "", [int(s_)], out=ns_bufs)
            ns_one.append(time.perf_counter() - t1)
    # the reference runs every list request's prefilter in a goroutine of its own (responsefilterer.go:165): three callers, each with result
    # buffers of its own, each step one batched walk -- their kernels fill each other's gaps
    conc = None
    if not only_batched:
        ncall = 3
        hbs = [eng.host_alloc(subs.size * words * 4 + subs.size * 8) for _ in range(ncall)]
        cb = [(b_[:subs.size * words * 4].view(np.uint32).reshape(subs.size, words), b_[subs.size * words * 4:].view(np.uint64)) for b_ in hbs]
        go = threading.Event()

        def caller(i):
            go.wait()
            for _ in range(steps):
                eng.lookup_ids_batch(rt, perm_name, st, "", subs, out=cb[i])

        for i in range(ncall):
            eng.lookup_ids_batch(rt, perm_name, st, "", subs, out=cb[i])
        ts = [threading.Thread(target=caller, args=(i,)) for i in range(ncall)]
        for t_ in ts:
            t_.start()
        tc = time.perf_counter()
        go.set()
        for t_ in ts:
            t_.join()
        elc = time.perf_counter() - tc
        conc = {"callers": ncall, "lookups_per_s": ncall * steps * subs.size / elc, "equal_to_sequential_run": bool(all(np.array_equal(c_[0], bms) and np.array_equal(c_[1], counts) for c_ in cb))}
        for b_ in hbs:
            eng.host_free(b_)
    eng.host_free(hb)
    if stats.get("rev_local_passes"):
        kname, kms, launches = "k_rev_local", stats["rev_local_ms"], max(1, stats["rev_local_passes"])
    else:
        kname, kms, launches = "k_rev_expand", stats["expand_ms"], max(1, stats["expand_launches"])
    lb = c3_lookup_bytes(w, subs)
    batch_bytes = float(lb.sum())
    ach = batch_bytes * steps / (kms * 1e-3) / 1e9 if kms > 0 else None
    traffic = traffic_bounds = None
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr):
        try:
            tj = json.load(open(tr))
            det = tj.get("C3_detail") or {}
            if tj.get("C3") and det.get("kernel") == kname:
                # counters -> bytes by the round-4 calibration (calibrated_traffic): the reverse walk gathers descriptors (counter exact) and reads
                # reverse rows as short coalesced runs (counter between exact and half): the midpoint of the two bounds, both recorded
                lo_b = float(det.get("fetch_bytes_per_launch_raw") or 0.0) + float(det.get("write_bytes_per_launch") or 0.0)
                hi_b = 2.0 * float(det.get("fetch_bytes_per_launch_raw") or 0.0) + float(det.get("write_bytes_per_launch") or 0.0)
                traffic = (lo_b + hi_b) / 2.0
                traffic_bounds = {"lower": lo_b, "upper": hi_b, "write_bytes": float(det.get("write_bytes_per_launch") or 0.0), "profile": det.get("tag"),
                                  "source": "profiles/traffic.json (an earlier run's rocprofv3 --pmc passes); midpoint of the calibrated bounds, profiles/r04_counter_calibration.md"}
        except Exception:  # noqa: BLE001
            pass
    out = {"metric": "lookup_resources_per_sec", "value": subs.size * steps / el, "unit": "lookups/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * el / steps, "workload": WORKLOAD_DESC["C3"], "lookups_per_step": int(subs.size), "relationships": w.ntuples,
           "objects": int(sum(w.nobjects.values())),
           "allowed_ids_per_lookup": float(np.mean(counts)), "allowed_ids_per_sec": float(np.sum(counts)) * steps / el,
           "p50_batch_ms": 1e3 * float(np.median(lat)), "p50_single_lookup_ms": 1e3 * float(np.median(one)), "p95_single_lookup_ms": 1e3 * float(np.percentile(one, 95)),
           "p50_single_namespace_lookup_ms": (1e3 * float(np.median(ns_one))) if ns_one else None,
           "pageable_result_buffers": {"p50_batch_ms": 1e3 * float(np.median(pg)), "lookups_per_s": subs.size / float(np.median(pg)), "equal_to_pinned_run": pageable_equal},
           "concurrent_callers": conc,
           "kernel_ms_per_step": stats["kernel_ms"] / steps, "launches_per_step": launches / steps, "reverse_levels": int(stats.get("levels_last", 0)),
           "bitmap_bytes_per_lookup": int(bms.shape[1] * 4),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
                        "traffic_detail": traffic_bounds, "kernel": kname, "kernel_avg_us": 1e3 * kms / launches,
                        "algorithmic_bytes_per_lookup": batch_bytes / subs.size, "algorithmic_bytes_per_launch": batch_bytes * steps / launches,
                        "model": "SURVEY.md 8(d) LookupResources formula on the generator's arrays: 17 + sum over reverse rows (8 + 4 deg) + N_pod / 8"}}
    if not pageable_equal or (conc and not conc["equal_to_sequential_run"]):
        out.setdefault("parity", {})["pageable_vs_pinned_mismatch" if not pageable_equal else "concurrent_vs_sequential_mismatch"] = True
    if not args.no_cpu:
        from oracle import orc
        o = orc.Oracle(w.schema)
        w.load(o)
        o.freeze()
        nt = min(usable_cores(), 32)
        npod = w.nobjects[rt]
        pods = np.arange(npod, dtype=np.uint32)
        mism = 0
        t1 = time.perf_counter()
        for i, s_ in enumerate(subs):  # the DEFINITION {id : Check == HAS} over every pod, multi-threaded bulk check: all 64 users
            op, _oe = o.check_bulk_ids_mt(nt, rt, perm_name, pods, st, "", np.full(npod, s_, dtype=np.uint32))
            want = np.flatnonzero(op == 2)
            got = np.flatnonzero(np.unpackbits(bms[i].view(np.uint8), bitorder="little"))
            mism += int(not np.array_equal(got, want))
        t_cpu = time.perf_counter() - t1
        # the CPU baseline proper: the same reverse walk on the host (numpy over CSR-by-subject rows); its id sets are checked too
        walks, t_walk = c3_cpu_reverse_walk(w, subs)
        for i in range(subs.size):
            got = np.flatnonzero(np.unpackbits(bms[i].view(np.uint8), bitorder="little"))
            mism += int(not np.array_equal(got, walks[i]))
        out["parity"] = {"lookups_checked_against_oracle": int(subs.size), "mismatches": mism + int(not pageable_equal) + int(bool(conc) and not conc["equal_to_sequential_run"]),
                         "checkers": "the DEFINITION {id : Check == HAS} over every pod (multi-threaded oracle) AND a CPU reverse walk"}
        # cpu_baseline = the oracle through its own ABI, as for the Check configs: the restated engine answers LookupResources by its DEFINITION
        # ({id : Check == HAS} over every pod, here split over the host threads).  The numpy reverse walk -- the algorithm the device runs,
        # specialised to C3's schema, one thread -- rides beside it as the tuned CPU figure.
        out["cpu_baseline"] = {"value": subs.size / t_cpu, "unit": "lookups/s", "cores": nt, "kind": "port", "seconds": round(t_cpu, 2),
                               "sample": f"all {subs.size} power users, restated CPU oracle (not embedded SpiceDB): {{id : Check == HAS}} over every pod = {subs.size * npod} oracle checks on {nt} threads",
                               "tuned_reverse_walk": {"value": subs.size / t_walk, "cores": 1, "seconds": round(t_walk, 3),
                                                      "sample": "reverse walk over CSR-by-subject rows in numpy (one thread; specialised to C3's schema; index building not timed)"}}
    return out


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

    native = args.native_loop == "on"  # the level loops inside libaclgpu.so (acl_shard_check_bulk / acl_shard_lookup_bulk) instead of the host-driven step protocol

    def run(se, barrier):
        e = se.shard.e
        se._alloc(1 << 20)
        items = e.make_items(rt, perm_name, w.res, st, "", w.subj)
        d_items = torch.from_numpy(items.view(np.uint8).copy()).to(se.shard.device)

        def check():
            if native:
                p_, e_, _s = se.check_bulk_ids_native(d_items)
                return p_, e_
            return se.check_bulk_ids(d_items)

        def lookup(sub):
            if native:
                return se.lookup_ids_batch_native(rt, perm_name, st, "", [sub])[0]
            return se.lookup_ids_batch(rt, perm_name, st, "", [sub])

        p, er = check()  # warm-up (also builds + uploads the shard's snapshot)
        bm = lookup(int(w.lookup_subjects[0]))
        barrier()
        t0 = time.perf_counter()
        tc = tf = 0.0
        for o in ops:
            t1 = time.perf_counter()
            if o == "C":
                p, er = check()
                tc += time.perf_counter() - t1
            else:
                bm = lookup(o[1])
                tf += time.perf_counter() - t1
        barrier()
        el = time.perf_counter() - t0
        st_ = e.stats()
        # cross-path property at full size: the last Filter bitmap must agree with sharded Checks of the same subject
        last_f = [o for o in ops if o != "C"][-1][1]
        rng = np.random.default_rng(7)
        pods = rng.integers(0, w.nobjects[rt], size=20000).astype(np.uint32)
        bm = lookup(last_f)
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
                          "level_loop": "inside libaclgpu.so (acl_shard_check_bulk / acl_shard_lookup_bulk: headers every level, entry blocks where the previous batch exported)" if native else "host-driven step protocol (aclgpu/sharded.py)",
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


SHARDED_CHECKPOINT = None  # set in the isolated child: called on rank 0 after every exchange form with the results so far


def isolated_sharded_leg(rank, world, timeout_s=180.0):
    """Runs the sharded leg of this very command line in CHILD processes (one per rank, a process group of their own on another port)
    and returns rank 0's result.  The leg drives RCCL paths that the one-GPU development boxes cannot exercise for real (grouped
    send/recv, the library's own communicator): whatever they do on a real 8-GPU node -- a crash, a wedged collective -- the parent
    still owns the measured headline line and prints it exactly once, with what the child got to before it died."""
    import signal
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"aclgpu_sharded_{os.getpid()}.json")
    errp = path + ".err"
    for f in (path, errp):
        if os.path.exists(f):
            os.remove(f)
    env = dict(os.environ)
    if world > 1:
        env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 101)
        env["TORCHELASTIC_USE_AGENT_STORE"] = "False"  # rank 0 of the children hosts the rendezvous store itself
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--sharded-child", path]
    note = None
    with open(errp, "w") as ef:
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=ef, start_new_session=True)
        try:
            pr.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(pr.pid, signal.SIGKILL)  # (the session this call started: nothing else is in it)
            pr.wait()
            note = f"sharded leg did not finish within {timeout_s:.0f} s (collective wedged?)"
    if rank != 0:
        return None
    res = {}
    if os.path.exists(path):
        try:
            res = json.load(open(path))
        except ValueError:
            res = {}
    if pr.returncode and not note:
        tail = open(errp).read().strip().splitlines()[-3:]
        note = f"sharded leg child exited with {pr.returncode}: " + " | ".join(tail)[-300:]
    if note:
        res["error"] = note
    res["isolated"] = "ran in child processes with a process group of their own"
    return res


def sharded_leg(args, w, replica, res, subj, world, rank, local_rank, result):
    """The north star's multi-GPU layout: rows partitioned by hash(object type) mod G (FNV-1a, avalanched), one shard per
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
    if args.native_loop == "on":
        modes = modes + ["native"]  # last: whatever it does on a real communicator, the host-driven forms' numbers are already out

    def run(se, comm_barrier, after_mode=None):
        d_items = torch.from_numpy(items.view(np.uint8).copy()).to(se.shard.device)
        res_by_mode = {}
        for mode in modes:
            if mode == "native":  # the whole level loop inside libaclgpu.so (acl_shard_check_bulk[_rccl]): one fixed-capacity all-gather per level
                try:
                    nstat = {}
                    for _ in range(2):
                        p, e, nstat = se.check_bulk_ids_native(d_items)
                    comm_barrier()
                    t0 = time.perf_counter()
                    lat = []
                    for _ in range(steps):
                        t1 = time.perf_counter()
                        p, e, nstat = se.check_bulk_ids_native(d_items)
                        lat.append(time.perf_counter() - t1)
                    comm_barrier()
                    el = time.perf_counter() - t0
                    mism = int((p.cpu().numpy() != want_p).sum() + (e.cpu().numpy() != want_e).sum())
                    res_by_mode[mode] = {"elapsed": el, "lat": lat, "mismatches": mism, "levels": nstat.get("levels"),
                                         "recv_entries_per_batch": nstat.get("entries_exchanged"), "exchanges_per_batch": nstat.get("exchanges"),
                                         "host_syncs_per_batch": nstat.get("host_syncs")}
                except Exception as ex:  # noqa: BLE001
                    res_by_mode[mode] = {"elapsed": float("inf"), "lat": [0.0], "mismatches": 0, "levels": None, "recv_entries_per_batch": None,
                                         "exchanges_per_batch": None, "error": f"{type(ex).__name__}: {ex}"}
                if after_mode:
                    after_mode(mode, res_by_mode[mode])
                continue
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
        "layout": f"hash(object type) mod {G} (FNV-1a, avalanched); 16 B frontier entries cross shards once per level",
        "transport": "RCCL over xGMI (torch.distributed nccl)" if world > 1 else "in-process copies between logical shards on ONE GPU (emulated, not a multi-GPU measurement)",
        "shards": G, "batch": n, "steps": steps})

    def mode_done(mode, ms):  # ms: every rank's record of one exchange form (filled in as soon as that form has run)
        el = max(m["elapsed"] for m in ms)
        if True:
            if any(m.get("error") for m in ms):
                result[mode] = {"error": next(m["error"] for m in ms if m.get("error"))}
                return
            result[mode] = {"collective": "all_gather_into_tensor of every export buffer, each rank keeps what it owns" if mode == "allgather"
                            else "exports grouped by owner on the device, batch_isend_irecv (grouped send/recv) to the owners only" if mode == "alltoall"
                            else "level loop inside libaclgpu.so: ONE fixed-capacity all-gather per level ([header | entries] per shard), decisions on the device, "
                                 "one host sync per burst of levels (ncclAllGather / ncclAllReduce from the library when world > 1)",
                            "decisions_per_s": n * steps / el, "ms_per_batch": 1e3 * el / steps, "p50_batch_ms": 1e3 * float(np.median(ms[0]["lat"])),
                            "levels": ms[0]["levels"], "exchanges_per_batch": ms[0]["exchanges_per_batch"],
                            "recv_entries_per_batch_by_shard": [m["recv_entries_per_batch"] for m in ms],
                            "mismatches_vs_replica": int(sum(m["mismatches"] for m in ms))}
        if "decisions_per_s" in result.get(modes[0], {}):
            result["decisions_per_s"] = result[modes[0]]["decisions_per_s"]
        result["mismatches_vs_replica"] = int(sum(result[m].get("mismatches_vs_replica", 0) for m in modes if m in result))

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
                if SHARDED_CHECKPOINT:
                    SHARDED_CHECKPOINT(result)

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


def single_checks_leg():
    """The proxy's own call shape (check.go:76-94, watch.go:50): T concurrent callers issuing single CheckPermission calls through the
    micro-batcher.  Native threads are needed to load it, so the leg runs tools/bin/batcher_bench (C++ over include/aclgpu.h, built by
    __graft_entry__.build()) on a 3-level pod graph of 500 k relationships and reports both call forms: one OS thread blocked per check
    (acl_check_one) and the completion queue (acl_check_one_submit / acl_check_completions: T logical callers over 8 OS threads)."""
    exe = os.path.join(ROOT, "tools", "bin", "batcher_bench")
    if not os.path.exists(exe):
        return {"skipped": "tools/bin/batcher_bench is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    try:
        pr = subprocess.run([exe, "1000", "64", "256", "1024"], capture_output=True, text=True, timeout=150)
    except Exception as ex:  # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}"}
    if pr.returncode:
        return {"error": f"batcher_bench exited {pr.returncode}: {pr.stderr.strip()[-200:]}"}
    txt = pr.stdout
    res = {"graph": "3-level pod graph, 500 k relationships; 1000 checks per caller", "unit": "checks/s", "blocking_threads": {}, "completion_queue": {}, "small_batch_p50_us": {}}
    for ln in txt.splitlines():
        try:
            r = json.loads(ln)
        except ValueError:
            continue
        if "small_batch_items" in r:
            res["small_batch_p50_us"][str(r["small_batch_items"])] = r["p50_us"]
        elif "logical_callers" in r:
            res["completion_queue"][str(r["logical_callers"])] = {"checks_per_s": r["checks_per_s"], "mean_latency_us": r["mean_latency_us"], "os_threads": r["threads"], "errors": r["errors"]}
        elif r.get("mode", "").startswith("micro-batched"):
            res["blocking_threads"][str(r["threads"])] = {"checks_per_s": r["checks_per_s"], "mean_latency_us": r["mean_latency_us"], "errors": r["errors"]}
    return res


def name_objects(eng, w):
    """Gives every object of the check stream's resource and subject types a NAME, in id order, BEFORE the numeric bulk load: the engine's dense ids are
    handed out in interning order, so name k gets id k and the bulk-loaded (anonymous-id) relationships are relationships of these named objects.
    pods are `ns<namespace>/pod-<id>` (the proxy's `{{namespacedName}}` shape, deploy/rules.yaml:68), users `user-<id>`."""
    rt, _p, st = w.check
    pod_ns = None
    for e_ in w.edges:
        if e_[0] == rt and e_[1] == "namespace":
            pod_ns = np.zeros(w.nobjects[rt], dtype=np.int64)
            pod_ns[e_[4]] = e_[5]
    names = {rt: [f"ns{int(pod_ns[i]) if pod_ns is not None else 0}/pod-{i}" for i in range(w.nobjects[rt])], st: [f"user-{i}" for i in range(w.nobjects[st])]}
    import ctypes
    out = ctypes.c_uint32()
    for t_, ns_ in names.items():
        tid = eng.type_id(t_)
        for k, nm in enumerate(ns_):
            eng._check(eng._L.acl_intern(eng._h, tid, nm.encode(), ctypes.byref(out)))
        if ns_ and out.value != len(ns_) - 1:
            raise SystemExit("name_objects must run on an empty engine (ids follow interning order)")
    w.names = names


def _local_edges(engine):
    return engine.stats().get("snapshot_edges_local", 0)


def check_bench(args, w, eng, steps, warmup, world, rank, label, legs, dist=None):
    """All Check legs of one workload on one engine.  Returns (record, gpu_perm, gpu_err) -- the answers of batch 0."""
    import torch
    rt, perm_name, st = w.check
    n = int(w.res.size)
    nrep = max(1, len(getattr(args, "replica_devices", []) or []))  # in-process replicas (--devices): callers and batches per step scale with them
    NB = max(8, args.callers * nrep + 2)  # distinct pre-generated batches: rotations of the request stream (same mix, different items in every lane)
    items0 = eng.make_items(rt, perm_name, w.res, st, "", w.subj)
    rec = {"workload": WORKLOAD_DESC[label], "batch": n, "relationships": w.ntuples, "objects": int(sum(w.nobjects.values()))}

    # ---------------- (i) device-resident, sequential, HIP events on: the roofline's kernel time
    d_batches = [torch.from_numpy(np.roll(items0, b * 4099).view(np.uint8).copy()).cuda() for b in range(NB)]
    d_perm = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_err = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(warmup):
        eng.check_bulk_ids_device(d_batches[k % NB].data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
    eng.stats_reset()
    eng.set_timing(True)
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for k in range(steps):
        t1 = time.perf_counter()
        eng.check_bulk_ids_device(d_batches[(k + 1) % NB].data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
        lat.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    el_dev = time.perf_counter() - t0
    eng.set_timing(False)
    st_dev = eng.stats()
    eng.check_bulk_ids_device(d_batches[0].data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
    gpu_perm, gpu_err = d_perm.cpu().numpy(), d_err.cpu().numpy()
    # the dominant kernel: the single-launch walk (k_check_local: every level of the batch in ONE launch) unless the batch took the
    # level loop (k_expand: one launch per level) -- frontier overflow, sharded graph, ACL_LOCAL_MAX=0
    if st_dev.get("local_ms", 0.0) >= st_dev["expand_ms"]:
        kern = {"name": "k_check_local", "ms": st_dev["local_ms"], "launches": max(1, st_dev["local_passes"])}
    else:
        kern = {"name": "k_expand", "ms": st_dev["expand_ms"], "launches": max(1, st_dev["expand_launches"])}
    rec["device_resident"] = {"decisions_per_s": n * steps / el_dev, "ms_per_batch": 1e3 * el_dev / steps, "p50_batch_ms": 1e3 * float(np.median(lat)),
                              "kernel_ms_per_batch": st_dev["kernel_ms"] / steps, "dominant_kernel": kern["name"],
                              "launches_per_batch": kern["launches"] / steps, "level_loop_launches_per_batch": st_dev["expand_launches"] / steps,
                              "note": "acl_check_bulk_ids_device: batch already in HBM, one call at a time"}
    rec["levels"] = int(st_dev["levels_last"])
    rec["has_fraction"] = float((gpu_perm == 2).mean())
    if legs != "device":
        # ... and the same device-resident calls from `callers` threads at once (own answer buffers each): launches of different batches overlap --
        # one's blocks move in as the other's finish -- which is what the host-id leg below gets too; the sequential figure above is what
        # prices the roofline (one launch at a time, HIP events)
        ncall = max(1, args.callers)
        outs_c = [(torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")) for _ in range(ncall)]
        go = threading.Event()
        warm = threading.Barrier(ncall + 1)

        def dev_caller(i):
            # (one untimed call each, all at once: every caller's evaluation context -- 1 GB of frontier buffers -- exists before the clock starts)
            eng.check_bulk_ids_device(d_batches[i % NB].data_ptr(), n, outs_c[i][0].data_ptr(), outs_c[i][1].data_ptr())
            warm.wait()
            go.wait()
            for k in range(i, steps, ncall):
                eng.check_bulk_ids_device(d_batches[k % NB].data_ptr(), n, outs_c[i][0].data_ptr(), outs_c[i][1].data_ptr())

        ts = [threading.Thread(target=dev_caller, args=(i,)) for i in range(ncall)]
        for t_ in ts:
            t_.start()
        warm.wait()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        go.set()
        for t_ in ts:
            t_.join()
        torch.cuda.synchronize()
        rec["device_resident"]["concurrent"] = {"callers": ncall, "decisions_per_s": n * steps / (time.perf_counter() - tc)}
        del outs_c
    if "device" == legs:
        rec["value"] = rec["device_resident"]["decisions_per_s"]
        rec["elapsed"] = el_dev
        rec["kernel"] = kern
        return rec, gpu_perm, gpu_err

    # ---------------- (ii) host ids, pipelined: THE timed region (`value`)
    hb = eng.host_alloc(NB * n * 21)
    h_items = hb[:NB * n * 16].view(aclgpu_item_dtype()).reshape(NB, n)
    h_perm = hb[NB * n * 16:NB * n * 17].reshape(NB, n)
    h_err = hb[NB * n * 17:].view(np.int32).reshape(NB, n)
    for b in range(NB):
        h_items[b] = np.roll(items0, b * 4099)
    window = max(1, min(args.window, NB))
    callers = max(1, args.callers) * nrep
    steps_all = steps * nrep  # a step = one batch per replica

    def submit_wait(k_steps):  # acl_check_bulk_ids_submit / acl_ticket_wait: `window` batches in flight from ONE host thread
        q = []
        for k in range(k_steps):
            if len(q) == window:
                eng.wait(q.pop(0))
            b = k % NB
            q.append(eng.submit_ids(h_items[b], h_perm[b], h_err[b]))
        for t in q:
            eng.wait(t)

    def blocking(k_steps):  # `callers` host threads, each in a blocking acl_check_bulk_ids (what goroutines behind the cgo shim do)
        """-> fire(): the caller threads exist and are parked before the clock starts (a server's callers are not created per request)"""
        if callers == 1:
            def fire1():
                for k in range(k_steps):
                    b = k % NB
                    eng.check_bulk_ids_into(h_items[b], h_perm[b], h_err[b])
            return fire1
        nxt = [0]
        lk = threading.Lock()
        go = [False]

        def run():
            # (the callers are on their marks when the clock starts: a yield loop, not an Event -- waking three sleeping Python threads costs
            #  0.1-0.3 ms, 3-5 % of a 20-step region that is 5 ms long, and is the harness's time, not the engine's)
            while not go[0]:
                time.sleep(0)
            while True:
                with lk:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= k_steps:
                    return
                b = k % NB
                eng.check_bulk_ids_into(h_items[b], h_perm[b], h_err[b])

        # (two callers never share a batch's output buffers at the same time: steps are handed out in order, NB >= callers)
        ts = [threading.Thread(target=run) for _ in range(callers)]
        for t in ts:
            t.start()

        def fire():
            go[0] = True
            for t in ts:
                t.join()
        return fire

    pipelined = (lambda k_: (lambda: submit_wait(k_))) if args.pipeline == "submit" else blocking
    # warm-up long enough to reach the steady state of the mode: the HIP runtime sets up its copy paths lazily -- the first time a 2nd, then a
    # 3rd host<->device copy is in flight at once, hipMemcpyAsync blocks ~7 ms (profiles/r03_submit_window_trace.txt); a `window`-ticket
    # warm-up left the second of them for the timed region (the "630 M/s at windows 3-4" of earlier runs was that one stall averaged over 40 steps)
    pipelined(max(warmup * nrep, 3 * window, 3 * callers, 12))()
    fire = pipelined(steps_all)
    # (no cyclic-GC pass of the interpreter inside the K-step region: the harness holds lists of a million name strings, and a full collection
    #  that walks them is milliseconds of a region that is 4-5 ms long)
    import gc
    gc.collect()
    gc.disable()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fire()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    host_ok = bool(np.array_equal(h_perm[0], gpu_perm) and np.array_equal(h_err[0], gpu_err))
    rec["value"] = n * steps_all / elapsed
    rec["elapsed"] = elapsed
    # the same leg over >= 200 steps (the driver's 20-step region is 5.6 ms): reported beside the K-step figure, never as `value`
    long_steps = max(200, steps) * nrep
    fire_long = pipelined(long_steps)
    torch.cuda.synchronize()
    tl = time.perf_counter()
    fire_long()
    torch.cuda.synchronize()
    el_long = time.perf_counter() - tl
    # ... and once more with the engine's HIP events ON (every caller's context records begin / end on ITS stream around its launch): what a
    # launch lasts INSIDE this leg, where the callers' launches overlap -- blocks of the next batch move in while the previous one's last blocks
    # finish -- and how much of consecutive launches overlaps (the roofline's kernel time is the SEQUENTIAL device leg's and
    # is longer than ms_per_step).  A pass of its own, after `value` was taken: events cost a few us per call, and timing switches the lone
    # caller's two-slice split off.
    fire_ev = pipelined(steps_all)
    eng.stats_reset()
    eng.set_timing(True)
    torch.cuda.synchronize()
    te = time.perf_counter()
    fire_ev()
    torch.cuda.synchronize()
    el_ev = time.perf_counter() - te
    eng.set_timing(False)
    st_ev = eng.stats()
    k_ms, k_n = (st_ev.get("local_ms", 0.0), st_ev["local_passes"]) if st_ev.get("local_ms", 0.0) >= st_ev["expand_ms"] else (st_ev["expand_ms"], st_ev["expand_launches"])
    rec["timed_leg_kernels"] = {"kernel_us_in_leg": 1e3 * k_ms / max(1, k_n), "launches": int(k_n), "ms_per_step_with_events": 1e3 * el_ev / steps_all * nrep,
                                "overlap_factor": (k_ms * 1e-3) / el_ev if el_ev > 0 else None,
                                "note": "the timed leg repeated with HIP events on the callers' streams: kernel_us_in_leg = mean launch duration while the callers' launches "
                                        "overlap (longer than a sequential launch: the blocks share the chip); overlap_factor = sum of launch durations / wall time of the "
                                        "pass = launches in flight on average -- ms_per_step = kernel_us_in_leg / overlap_factor"}
    rec["host_ids"] = {"decisions_per_s": n * steps_all / elapsed, "ms_per_batch": 1e3 * elapsed / steps_all, "distinct_batches": NB, "answers_equal_device_leg": host_ok,
                       "long_run": {"steps": long_steps // nrep, "decisions_per_s": n * long_steps / el_long, "ms_per_step": 1e3 * el_long / (long_steps // nrep)},
                       "replicas_in_process": nrep, "replica_calls": eng.replica_calls(),
                       "mode": (f"acl_check_bulk_ids_submit/wait, {window} in flight" if args.pipeline == "submit" else f"{callers} caller thread(s) in blocking acl_check_bulk_ids"),
                       "note": "pinned host buffers in, pinned host buffers out: the items and the answers cross PCIe inside every call -- read and written by the kernel itself where the single-launch walk takes the batch (no copies), H2D + kernels + D2H otherwise"}
    # ---------------- batch latency: >= 200 single unpipelined host-id calls (SURVEY.md 8(d) "p50 over >= 200 batches after 20 warm-ups")
    nl = max(200, steps)
    lat = []
    for k in range(20 + nl):
        b = k % NB
        t1 = time.perf_counter()
        eng.check_bulk_ids_into(h_items[b], h_perm[b], h_err[b])
        if k >= 20:
            lat.append(time.perf_counter() - t1)
    rec["latency"] = {"p50_batch_ms": 1e3 * float(np.median(lat)), "p95_batch_ms": 1e3 * float(np.percentile(lat, 95)), "batches": nl,
                      "decisions_per_s_unpipelined": n / float(np.median(lat)), "note": "one acl_check_bulk_ids call at a time, pinned buffers (PCIe inclusive)"}
    pg = [time.perf_counter()]
    for _ in range(5):
        eng.check_bulk_ids(items0)
        pg.append(time.perf_counter())
    rec["latency"]["pageable_buffers_p50_ms"] = 1e3 * float(np.median(np.diff(pg)))
    # every one of the NB rotations has been answered by now (timed leg + 220 latency calls): batch b must hold batch 0's answers rotated
    # the same way -- cpu_and_roofline compares batch 0 with the oracle item by item, so this extends the oracle check to all NB batches
    rot_bad = 0
    for b in range(NB):
        rot_bad += int((h_perm[b] != np.roll(gpu_perm, b * 4099)).sum() + (h_err[b] != np.roll(gpu_err, b * 4099)).sum())
    rec["rotations"] = {"batches": NB, "items": NB * n, "mismatches_vs_rotated_batch0": rot_bad}
    eng.host_free(hb)
    # ---------------- (iii) string path on NAMED objects: every pod and user of the graph was given a name before the load (name_objects), so the strings
    # below resolve to the very ids of the batch and the answers must equal the id path's (which the oracle checks item by item).  Timed: the whole
    # ABI call -- strings -> ids on the host (parallel, straight into pinned staging), one device pass, answers back.  Two item forms: NUL-terminated
    # fields (acl_check_bulk) and {pointer, length} fields (acl_check_bulk_v: what a cgo shim points at Go strings without copying).
    names = getattr(w, "names", None)
    if names:
        # (the interpreter's cyclic collector off for these legs: the harness holds a million name strings, and a full collection in the middle of a 0.15 ms call was a
        #  9 ms straggler -- one of 40 calls, 60 % of that leg's mean rate in one of the round's runs)
        import gc
        gc_was = gc.isenabled()
        gc.disable()
        sp = {"note": "acl_check_bulk_v / acl_check_bulk on named objects (every pod and user of the graph has a name; tables of "
                      f"{len(names[rt])} + {len(names[st])} names); answers compared with the id path's", "sizes": {}}
        ok_all = True
        for m in (65536, 16384, 1024):  # (largest first: the pool's threads exist from the first call of 4 096 items on, as in a proxy that has served one list)
            m = min(m, n)
            qs = [(rt, names[rt][int(r_)], perm_name, st, names[st][int(s_)], "") for r_, s_ in zip(w.res[:m], w.subj[:m])]
            pv, pc, pp = eng.make_check_views(qs), eng.make_check_strings_named(qs), eng.make_check_packed(qs)
            row = {}
            for form, call, prep in (("views", eng.check_bulk_views, pv), ("c_strings", eng.check_bulk_prepared, pc), ("packed", eng.check_bulk_packed, pp)):
                got = call(prep)
                ok = bool(np.array_equal(got[0], gpu_perm[:m]) and np.array_equal(got[1], gpu_err[:m]))
                ok_all = ok_all and ok
                ts = []
                for _ in range(40):
                    t1 = time.perf_counter()
                    call(prep)
                    ts.append(time.perf_counter() - t1)
                row[form] = {"decisions_per_s": m / float(np.mean(ts)), "ms_per_batch": 1e3 * float(np.mean(ts)), "p50_ms": 1e3 * float(np.median(ts)), "best_ms": 1e3 * min(ts),
                             "calls": len(ts), "answers_equal_id_path": ok}  # (the rate is the MEAN over the calls, stragglers included)
                if form == "views":
                    # ... and 2 ms apart: the interning pool's workers poll for 150 us after a batch and are asleep by then -- what a call finds when the proxy is
                    # not busy (the back-to-back figure above is what it finds when it is)
                    ts = []
                    for _ in range(20):
                        time.sleep(0.002)
                        t1 = time.perf_counter()
                        call(prep)
                        ts.append(time.perf_counter() - t1)
                    row["views_2ms_apart"] = {"decisions_per_s": m / float(np.mean(ts)), "ms_per_batch": 1e3 * float(np.mean(ts)), "p50_ms": 1e3 * float(np.median(ts)), "calls": len(ts)}
            sp["sizes"][str(m)] = row
        # PostFilter's own shape (postfilter.go:67-119: K list items, ONE subject for every pair): acl_check_bulk_keep_v / _packed answer it by one reverse walk
        # from the subject + K bit tests (engine.cpp keep_by_reverse_walk); the keep mask is compared with the id path's answers for the same pairs
        kr = {"note": "K list items x 1 template for ONE user: acl_check_bulk_keep_v / acl_check_bulk_keep_packed (one reverse walk + K name tests) against the forward "
                      "string path (acl_check_bulk_v + the AND); keep masks compared with the id path's answers", "sizes": {}}
        # ... and for a user as the proxy's users are -- a handful of grants, not the benchmark graph's deep-group members who see a quarter of all pods:
        # `user-sparse` is made a direct viewer of 24 pods of the list (a write through the ABI: patched into the snapshot like any other)
        import aclgpu as aclgpu_mod
        grants = sorted({int(x) for x in w.res[:min(65536, n):2731]})[:24]
        eng.write([(aclgpu_mod.OP_TOUCH, (rt, names[rt][g_], "viewer", st, "user-sparse", "")) for g_ in grants])
        for m in (1024, 16384, 65536):
            m = min(m, n)
            rowk = {}
            for who, u_ in (("batch_user", int(w.subj[0])), ("other_user", int(w.subj[m // 2])), ("sparse_user", None)):
                uname_ = "user-sparse" if u_ is None else names[st][u_]
                qk = [(rt, names[rt][int(r_)], perm_name, st, uname_, "") for r_ in w.res[:m]]
                off = np.arange(m + 1, dtype=np.uint32)
                if u_ is None:
                    want = np.isin(w.res[:m], np.asarray(grants, dtype=w.res.dtype))
                else:
                    tp, te = eng.check_bulk_ids(eng.make_items(rt, perm_name, w.res[:m], st, "", np.full(m, u_, dtype=np.uint32)))
                    want = (tp == 2) & (te == 0)
                kv_prep, kp_prep = eng.make_check_views(qk), eng.make_check_packed(qk)
                before = eng.stats()["keep_route_calls"]
                entry = {"kept": int(want.sum())}
                for form, call, prep in (("keep_v", eng.check_bulk_keep_views, kv_prep), ("keep_packed", eng.check_bulk_keep_packed, kp_prep)):
                    got = call(prep, off).astype(bool)
                    okk = bool(np.array_equal(got, want))
                    ok_all = ok_all and okk
                    ts = []
                    for _ in range(40):
                        t1 = time.perf_counter()
                        call(prep, off)
                        ts.append(time.perf_counter() - t1)
                    entry[form] = {"items_per_s": m / float(np.mean(ts)), "ms_per_call": 1e3 * float(np.mean(ts)), "p50_ms": 1e3 * float(np.median(ts)), "mask_equals_id_path": okk}
                entry["calls_answered_by_the_reverse_walk"] = int(eng.stats()["keep_route_calls"] - before)
                # CheckBulkPermissions ITSELF of the same pairs (what filterItemsWithBulkPermissions sends, postfilter.go:134): the pair form of the same route --
                # on a recursive permission (C4's nested groups) once a forward sweep has shown that no Check of it ends at the depth limit on this snapshot
                # (engine.cpp no_object_is_deep: the first call at a snapshot goes forward, the second sweeps); permissionships AND errors compared with the id path's
                before = eng.stats()["keep_route_calls"]
                for _ in range(3):
                    gp, ge = eng.check_bulk_views(kv_prep)
                okp = bool(np.array_equal((gp == 2) & (ge == 0), want) and not ge.any())
                ok_all = ok_all and okp
                ts = []
                for _ in range(40):
                    t1 = time.perf_counter()
                    eng.check_bulk_views(kv_prep)
                    ts.append(time.perf_counter() - t1)
                entry["pairs_views"] = {"decisions_per_s": m / float(np.mean(ts)), "ms_per_call": 1e3 * float(np.mean(ts)), "p50_ms": 1e3 * float(np.median(ts)), "pairs_equal_id_path": okp,
                                        "calls_answered_by_the_reverse_walk": int(eng.stats()["keep_route_calls"] - before), "of_calls": 43}
                # (the forward string path at this size: the same number of pairs with mixed subjects, measured above)
                entry["forward_views"] = {"items_per_s": sp["sizes"][str(m)]["views"]["decisions_per_s"], "ms_per_call": sp["sizes"][str(m)]["views"]["ms_per_batch"]}
                rowk[who] = entry
            kr["sizes"][str(m)] = rowk
        sp["keep_route"] = kr
        big = sp["sizes"][str(min(65536, n))]
        sp.update({"decisions_per_s": big["views"]["decisions_per_s"], "ms_per_batch": big["views"]["ms_per_batch"], "items": min(65536, n), "answers_equal_id_path": ok_all,
                   "keep_route_items_per_s": {who_: e_["keep_v"]["items_per_s"] for who_, e_ in kr["sizes"][str(min(65536, n))].items()},
                   "one_user_pairs_decisions_per_s": {who_: e_["pairs_views"]["decisions_per_s"] for who_, e_ in kr["sizes"][str(min(65536, n))].items()},
                   "depth_sweeps": int(eng.stats()["depth_sweeps"]),
                   "packed_decisions_per_s": {k_: v_["packed"]["decisions_per_s"] for k_, v_ in sp["sizes"].items()},
                   "views_decisions_per_s": {k_: v_["views"]["decisions_per_s"] for k_, v_ in sp["sizes"].items()},
                   "views_2ms_apart_decisions_per_s": {k_: v_["views_2ms_apart"]["decisions_per_s"] for k_, v_ in sp["sizes"].items()}})
        rec["string_path"] = sp
        if gc_was:
            gc.enable()
        if not ok_all:
            rec["string_path_mismatch"] = True
    else:
        rec["string_path"] = {"skipped": "objects of this run have no names (--legs / workload without name_objects)"}
    rec["kernel"] = kern
    return rec, gpu_perm, gpu_err


def postfilter_c3_leg(local_rank):
    """The reference's PostFilter as the UNPATCHED proxy sends it (pkg/authz/postfilter.go:67-134): one CheckBulkPermissions of K pairs that all name the requesting
    user, on C3's graph -- a schema without recursion, as the reference's own bootstrap schema is, so that no Check of pod#view can end at the depth limit and the
    engine may answer the call by ONE reverse walk + bit tests (engine.cpp keep_by_reverse_walk, pair form).  Per kind of user: acl_check_bulk_v of 65 536 named
    pairs (mean of 30 back-to-back calls), the same pairs through names -> ids -> acl_check_bulk_ids (the forward walk; its id resolution is not timed), and
    whether every permissionship and error of the two agree."""
    import aclgpu
    from aclgpu import workloads
    w = workloads.c3(batch=65536)
    eng = aclgpu.Engine(w.schema, device=local_rank, eager_contexts=True)
    name_objects(eng, w)
    w.load(eng)
    eng.snapshot()
    rt, perm_name, st = w.check
    names = w.names
    m = min(65536, len(w.res))
    out = {"pairs": m, "note": "acl_check_bulk_v of one user's pairs (CheckBulkPermissions as filterItemsWithBulkPermissions sends it) on C3's named graph: by the reverse walk; "
                               "forward = the same pairs by id through acl_check_bulk_ids", "users": {}}
    ok_all = True
    for who, u in (("power_user", int(w.lookup_subjects[0])), ("ordinary_user", 3)):
        qk = [(rt, names[rt][int(r_)], perm_name, st, names[st][u], "") for r_ in w.res[:m]]
        prep = eng.make_check_views(qk)
        items16 = eng.make_items(rt, perm_name, w.res[:m], st, "", np.full(m, u, dtype=np.uint32))
        fp, fe = eng.check_bulk_ids(items16)
        before = eng.stats()["keep_route_calls"]
        gp, ge = eng.check_bulk_views(prep)
        same = bool(np.array_equal(gp, fp) and np.array_equal(ge, fe))
        ok_all = ok_all and same
        ts, tf = [], []
        for _ in range(30):
            t1 = time.perf_counter()
            eng.check_bulk_views(prep)
            ts.append(time.perf_counter() - t1)
        for _ in range(30):
            t1 = time.perf_counter()
            eng.check_bulk_ids(items16)
            tf.append(time.perf_counter() - t1)
        out["users"][who] = {"allowed_among_the_pairs": int((fp == 2).sum()), "decisions_per_s": m / float(np.mean(ts)), "p50_ms": 1e3 * float(np.median(ts)),
                             "calls_answered_by_the_reverse_walk": int(eng.stats()["keep_route_calls"] - before), "forward_by_id_decisions_per_s": m / float(np.mean(tf)),
                             "forward_by_id_p50_ms": 1e3 * float(np.median(tf)), "answers_equal_forward": same}
    out["answers_equal_forward"] = ok_all
    out["list_level"] = list_level_leg(eng, w, 10000)
    eng.close()
    return out


def pod_json(full_name, k):
    """A pod as a kube list response carries it, trimmed to ~2 KB: metadata with labels / annotations / ownerReferences / managedFields, a container, status."""
    ns, name = full_name.split("/", 1)
    return {"apiVersion": "v1", "kind": "Pod",
            "metadata": {"annotations": {"kubernetes.io/config.seen": "2026-01-01T00:00:00.000000000Z", "checksum/config": "%064x" % (k * 2654435761)},
                         "creationTimestamp": "2026-01-01T00:00:00Z", "generateName": name.rsplit("-", 1)[0] + "-",
                         "labels": {"app": "web", "pod-template-hash": "%010x" % k, "tier": "frontend \\ \"quoted\""}, "name": name, "namespace": ns,
                         "ownerReferences": [{"apiVersion": "apps/v1", "blockOwnerDeletion": True, "controller": True, "kind": "ReplicaSet", "name": "web-%x" % k,
                                              "uid": "00000000-0000-4000-8000-%012x" % k}],
                         "managedFields": [{"apiVersion": "v1", "fieldsType": "FieldsV1", "fieldsV1": {"f:metadata": {"f:labels": {".": {}, "f:app": {}}}, "f:spec": {"f:containers": {}}},
                                            "manager": "kube-controller-manager", "operation": "Update", "time": "2026-01-01T00:00:00Z"}],
                         "resourceVersion": str(1000000 + k), "uid": "11111111-0000-4000-8000-%012x" % k},
            "spec": {"containers": [{"image": "registry.example/web:1.%d" % (k % 50), "imagePullPolicy": "IfNotPresent", "name": "web",
                                     "ports": [{"containerPort": 8080, "protocol": "TCP"}], "resources": {"limits": {"cpu": "500m", "memory": "256Mi"}, "requests": {"cpu": "100m", "memory": "128Mi"}},
                                     "env": [{"name": "POD_NAME", "valueFrom": {"fieldRef": {"apiVersion": "v1", "fieldPath": "metadata.name"}}}],
                                     "terminationMessagePath": "/dev/termination-log", "terminationMessagePolicy": "File"}],
                     "dnsPolicy": "ClusterFirst", "nodeName": "node-%d" % (k % 97), "restartPolicy": "Always", "schedulerName": "default-scheduler", "serviceAccountName": "default",
                     "tolerations": [{"effect": "NoExecute", "key": "node.kubernetes.io/not-ready", "operator": "Exists", "tolerationSeconds": 300}]},
            "status": {"conditions": [{"lastTransitionTime": "2026-01-01T00:00:05Z", "status": "True", "type": t} for t in ("Initialized", "Ready", "ContainersReady", "PodScheduled")],
                       "hostIP": "10.0.%d.%d" % (k % 250, k % 199), "phase": "Running", "podIP": "10.244.%d.%d" % (k % 250, k % 251), "qosClass": "Burstable", "startTime": "2026-01-01T00:00:01Z"}}


def list_level_leg(eng, w, m, calls=15):
    """The filters on the BYTES of a kube list response: acl_filter_list_response = filterListResponse (postfilter.go:17-55: decode the
    list, K checks for the requesting user, re-encode) and acl_prefilter_response = filterList (responsefilterer.go:376-400) over a LookupResources bitmap, for a
    PodList of m pods (~2 KB each) of the named graph and its first power user.  The engine finds the items' spans with all host threads, validates and resolves
    them in parallel and splices the kept ones' original bytes (csrc/engine_list.cpp, json_index.hpp); generic_*: python's json (C, one thread) decoding and
    re-encoding the same body, which is what the reference does around its one CheckBulkPermissions with encoding/json."""
    import ctypes as C
    import json as js
    rt, perm_name, st = w.check
    names = w.names
    m = min(m, len(w.res))
    res = w.res[:m]
    u = int(w.lookup_subjects[0]) if getattr(w, "lookup_subjects", None) is not None else int(w.subj[0])
    uname = names[st][u]
    body = js.dumps({"apiVersion": "v1", "kind": "PodList", "metadata": {"resourceVersion": "123456"}, "items": [pod_json(names[rt][int(r_)], k) for k, r_ in enumerate(res)]},
                    separators=(",", ":")).encode()
    t1 = time.perf_counter()
    doc = js.loads(body)
    t_dec = time.perf_counter() - t1
    t1 = time.perf_counter()
    js.dumps(doc, separators=(",", ":"))
    t_enc = time.perf_counter() - t1
    del doc
    tp, te = eng.check_bulk_ids(eng.make_items(rt, perm_name, res, st, "", np.full(m, u, dtype=np.uint32)))
    want = (tp == 2) & (te == 0)
    template = f"{rt}:{{{{namespacedName}}}}#{perm_name}@{st}:{{{{user.name}}}}"
    out, kept, total = eng.filter_list_response(body, [template], uname)
    kept_names = [it["metadata"]["namespace"] + "/" + it["metadata"]["name"] for it in (js.loads(out)["items"] or [])]
    ok = bool(kept == int(want.sum()) and total == m and kept_names == [names[rt][int(r_)] for r_, k_ in zip(res, want) if k_])
    arr = (C.c_char_p * 1)(template.encode())
    outp, outn, k_, t_ = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
    ts = []
    for _ in range(calls):  # (the C call: the Python mirror's copy of the output body is the harness's)
        t1 = time.perf_counter()
        rc = eng._L.acl_filter_list_response(eng._h, body, len(body), arr, 1, uname.encode(), C.byref(outp), C.byref(outn), C.byref(k_), C.byref(t_))
        ts.append(time.perf_counter() - t1)
        eng._L.acl_free(outp)
        if rc:
            raise SystemExit("acl_filter_list_response failed")
    bms, _cnt = eng.lookup_ids_batch(rt, perm_name, st, "", [u])
    bm = np.ascontiguousarray(bms[0], dtype=np.uint32)
    out2, kept2, _total2 = eng.prefilter_response(rt, bm, "{{namespacedName}}", eng.BODY_LIST, body)
    tq = []
    for _ in range(calls):
        t1 = time.perf_counter()
        rc = eng._L.acl_prefilter_response(eng._h, eng.type_id(rt), bm.ctypes.data, bm.size, b"{{namespacedName}}", eng.BODY_LIST, body, len(body), C.byref(outp), C.byref(outn),
                                           C.byref(k_), C.byref(t_))
        tq.append(time.perf_counter() - t1)
        eng._L.acl_free(outp)
        if rc:
            raise SystemExit("acl_prefilter_response failed")
    return {"items": m, "body_MB": len(body) / 1e6, "kept": int(want.sum()), "user": "power user",
            "postfilter": {"p50_ms": 1e3 * float(np.median(ts)), "items_per_s": m / float(np.median(ts)), "body_GB_per_s": len(body) / float(np.median(ts)) / 1e9, "kept_equal_id_path": ok},
            "prefilter": {"p50_ms": 1e3 * float(np.median(tq)), "items_per_s": m / float(np.median(tq)), "body_GB_per_s": len(body) / float(np.median(tq)) / 1e9,
                          "body_equal_postfilter": bool(out2 == out and kept2 == kept)},
            "generic_decode_ms": 1e3 * t_dec, "generic_encode_ms": 1e3 * t_enc,
            "note": "bytes of a PodList in, filtered bytes out (postfilter.go:17-55 / responsefilterer.go:376-400); generic_* = python json.loads / json.dumps of the same body, one thread"}


def aclgpu_item_dtype():
    import aclgpu
    return aclgpu.ITEM_DTYPE


def cpu_and_roofline(args, w, rec, gpu_perm, gpu_err, label, with_oracle=None):
    """CPU oracle on the same batch (parity + baseline) and the roofline from the oracle's byte model over the WHOLE batch.
    with_oracle(o, cores): a caller's own checks against the loaded oracle (the C5R leg's lookup definition sample) before it is dropped."""
    from oracle import orc
    rt, perm_name, st = w.check
    n = int(w.res.size)
    t0 = time.time()
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    t_oload = time.time() - t0
    # single thread: calibrate, then a bounded sample (prefix of the SAME batch)
    m0 = min(n, 512)
    t0 = time.perf_counter()
    o.check_bulk_ids(rt, perm_name, w.res[:m0], st, "", w.subj[:m0])
    per = (time.perf_counter() - t0) / m0
    m = int(min(n, max(m0, min(args.cpu_seconds, 4.0) / max(per, 1e-9))))
    t0 = time.perf_counter()
    operm, oerr = o.check_bulk_ids(rt, perm_name, w.res[:m], st, "", w.subj[:m])
    t_cpu = time.perf_counter() - t0
    mism = int((operm != gpu_perm[:m]).sum() + (oerr != gpu_err[:m]).sum())
    # all host cores: the whole batch split statically over threads (SURVEY.md 8(d) "CPU baseline beside it" (b));
    # the box may expose more hardware threads than this container may use: keep the thread count that is fastest
    cores = usable_cores()
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
    t0 = time.perf_counter()
    mperm, merr = o.check_bulk_ids_mt(cores, rt, perm_name, w.res, st, "", w.subj)  # EVERY answer of the batch is checked
    t_mt = time.perf_counter() - t0
    mism_mt = int((mperm != gpu_perm).sum() + (merr != gpu_err).sum())
    # ... and the TUNED CPU Check beside it (oracle/acl_oracle.c orc_tuned_*: a row index instead of three binary searches per row, the
    # level-synchronous frontier the device walks with identical states merged, dynamic chunks over the same host threads): the same batch, every answer
    # asserted equal to the recursive evaluator's.  Building its index is setup (like the snapshot build on the other side) and timed apart.
    tuned = None
    t0 = time.perf_counter()
    if o.tuned_build():
        t_tb = time.perf_counter() - t0
        tt = []
        for _ in range(2):
            t0 = time.perf_counter()
            tperm, terr = o.tuned_check_bulk_ids_mt(cores, rt, perm_name, w.res, st, "", w.subj)
            tt.append(time.perf_counter() - t0)
        tuned = {"value": n / min(tt), "unit": "decisions/s", "cores": cores, "seconds": round(min(tt), 3), "index_build_s": round(t_tb, 2),
                 "equal_to_recursive_oracle": bool(np.array_equal(tperm, mperm) and np.array_equal(terr, merr)),
                 "sample": f"the whole {n}-item batch, {cores} host threads, best of 2; oracle/acl_oracle.c orc_tuned_check_bulk_ids_mt (checker-side code: never linked into the product)"}
        mism_mt += int((tperm != mperm).sum() + (terr != merr).sum())
    if with_oracle is not None:
        with_oracle(o, cores)
    rot = rec.pop("rotations", None)
    str_bad = int(bool(rec.pop("string_path_mismatch", False)))  # named-object strings answered differently from the ids they name
    rec["parity"] = {"checked_against_oracle": n, "mismatches": mism + mism_mt + str_bad}
    if rot:  # all 8 distinct batches of the timed leg: batch b == oracle answers rotated by b * 4099 (the oracle's answers equal batch 0's item by item)
        rec["parity"].update({"checked_against_oracle": rot["items"], "distinct_batches_checked": rot["batches"],
                              "mismatches": mism + mism_mt + rot["mismatches_vs_rotated_batch0"]})
    rec["cpu_baseline"] = {"value": n / t_mt, "unit": "decisions/s", "cores": cores, "kind": "port",
                           "sample": f"the whole {n}-item batch split statically over {cores} host threads, restated CPU oracle (not embedded SpiceDB)",
                           "seconds": round(t_mt, 2), "load_s": round(t_oload, 2),
                           "single_thread": {"value": m / t_cpu, "sample": f"first {m} items", "seconds": round(t_cpu, 2)},
                           "tuned": tuned}
    # algorithmic bytes (SURVEY.md 8(d) model) over the WHOLE batch, multi-threaded, with the per-level split
    t0 = time.perf_counter()
    tot, lvl_b, lvl_s = o.check_bytes_bulk(max(cores, 4), rt, perm_name, w.res, st, "", w.subj)
    t_bytes = time.perf_counter() - t0
    k = rec.pop("kernel")
    steps = rec["_steps"]
    ach = tot * steps / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else None
    nz = int(np.flatnonzero(lvl_b)[-1]) + 1 if lvl_b.any() else 1
    traffic = traffic_detail = None  # `traffic`: HBM bytes per launch (a number, as the bench contract defines it); the label rides beside it
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr):
        try:
            tj = json.load(open(tr))
            det = tj.get(label + "_detail") or {}
            if tj.get(label) and det.get("kernel") == k["name"]:  # (only a profile of the SAME kernel says anything about this run's launches)
                traffic_detail = calibrated_traffic(float(det.get("fetch_bytes_per_launch_raw") or 0.0), float(det.get("write_bytes_per_launch") or 0.0), n)
                traffic = float(traffic_detail["bytes_per_launch"])
                traffic_detail.update({"profile": det.get("tag"),
                                       "source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command, NOT measured in "
                                                 "this run; counters -> bytes by the round-4 calibration"})
        except Exception:  # noqa: BLE001
            pass
    if getattr(args, "traffic", "static") == "measure":
        m = measure_traffic(args, label, k["name"], n)
        if m.get("bytes_per_launch"):
            traffic, traffic_detail = float(m["bytes_per_launch"]), m
        elif traffic_detail is not None:
            traffic_detail["live_measurement_failed"] = m.get("error")
    k_s = k["ms"] * 1e-3 / k["launches"] if k.get("launches") else 0.0  # one launch, seconds
    rec["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
                       # HBM UTILISATION, as opposed to the byte model's `frac`: counter traffic of a launch / its duration / peak
                       "traffic_gbs": (traffic / k_s / 1e9) if (traffic and k_s) else None, "traffic_frac": (traffic / k_s / 1e9 / HBM_PEAK_GBS) if (traffic and k_s) else None,
                       "traffic_detail": traffic_detail, "kernel": k["name"], "kernel_avg_us": 1e3 * k["ms"] / k["launches"], "launches_per_batch": k["launches"] / steps,
                       "measured_in": "device_resident leg (sequential launches, HIP events on the launching stream)",
                       "algorithmic_bytes_per_check": tot / n, "algorithmic_bytes_per_batch": tot, "algorithmic_bytes_per_launch": tot * steps / k["launches"],
                       "algorithmic_bytes_by_model_level": [int(x) for x in lvl_b[1:nz]], "distinct_states_by_model_level": [int(x) for x in lvl_s[1:nz]],
                       "model": "oracle byte counter over all items (level-synchronous, sorted-row probes; the model's levels count every computed "
                                "userset as a dispatch, the kernels inline them)", "model_seconds": round(t_bytes, 2)}


def calibrated_traffic(fetch_raw, write_bytes, n_items, answer_bytes_per_item=6):
    """HBM bytes of one launch from its FETCH_SIZE / WRITE_SIZE counter values, by the calibration of profiles/r04_counter_calibration.md
    (tools/counter_calib.hip: every access shape of this engine at a known byte count, same rocprofv3 passes):
      * FETCH_SIZE counts 64 B per read REQUEST whatever its size (TCC_BUBBLE, the 128-B-request counter, reads 0 on gfx950): exact for the
        4 / 8 / 16 B gathers (each fetches one 64 B sector), HALF for coalesced streams (128 B requests) -- the guide's x2 applies to those only;
      * WRITE_SIZE is exact for coalesced and wave-compacted stores (32 B granules).
    The walk's only coalesced reads are its own frontier (every entry it wrote is read back once, 64 lanes x 16 B side by side) and the items;
    so reads = counter + (frontier bytes written + items) / 2, frontier bytes written = WRITE_SIZE - answers.  The blanket x2 of rounds 1-3
    (an upper bound) and the raw sum (a lower bound) ride beside the estimate."""
    stream = max(0.0, write_bytes - n_items * answer_bytes_per_item) + n_items * 16.0
    stream = min(stream, 2.0 * fetch_raw)  # (the stream part cannot exceed what the counter saw at half weight)
    return {"bytes_per_launch": fetch_raw + stream / 2.0 + write_bytes, "fetch_bytes_raw": fetch_raw, "write_bytes": write_bytes,
            "stream_read_bytes_estimated": stream, "bytes_per_launch_lower_bound": fetch_raw + write_bytes, "bytes_per_launch_upper_bound": 2.0 * fetch_raw + write_bytes,
            "calibration": "profiles/r04_counter_calibration.md: FETCH_SIZE = 64 B x read requests (exact for gathers, half for 128 B stream requests); WRITE_SIZE exact"}


def measure_traffic(args, label, kernel, n_items=0):
    """HBM traffic of `kernel` per launch, measured NOW: this command's device-resident leg re-run twice under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE and WRITE_SIZE in separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; no other trace domain), mean over the
    child's launches, turned into bytes by calibrated_traffic() (round 4: the calibration on this engine's own access shapes replaces the
    blanket x2 of the guide's stream case).  {"error": ...} when rocprofv3 is missing or a pass fails -- the caller then keeps the figure of
    profiles/traffic.json, labelled as such."""
    import glob
    import shutil
    import signal
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {"error": "rocprofv3 not found"}
    child = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--scale", str(args.scale), "--steps", "3", "--warmup", "1", "--no-cpu", "--legs", "device",
             "--configs", "off", "--strings", "off", "--traffic", "static"]
    if label == "C2":
        child[child.index("--workload") + 1] = "C2"
    if args.batch and label != "C2":
        child += ["--batch", str(args.batch)]
    if label == "C5R":
        child += ["--replica"]
    per = {}
    t0 = time.time()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="aclgpu_pmc_")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k_, None)
            pr = subprocess.Popen([exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "r", "--"] + child, cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)  # (the session this call started: nothing else is in it)
                pr.wait()
                return {"error": f"rocprofv3 --pmc {ctr} pass did not finish within 150 s"}
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if pr.returncode or not dbs:
                return {"error": f"rocprofv3 --pmc {ctr} pass failed (rc {pr.returncode})"}
            con = sqlite3.connect(dbs[0])
            try:
                rows = con.execute("select value from counters_collection where counter_name=? and kernel_name like ?", (ctr, f"%{kernel}%")).fetchall()
            finally:
                con.close()
            if not rows:
                return {"error": f"no {ctr} samples of {kernel}"}
            per[ctr] = (sum(r[0] for r in rows) / len(rows) * 1024.0, len(rows))  # KiB -> bytes
        except Exception as ex:  # noqa: BLE001
            return {"error": f"{type(ex).__name__}: {ex}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f, w_ = per["FETCH_SIZE"][0], per["WRITE_SIZE"][0]
    out = calibrated_traffic(f, w_, n_items)
    out.update({"launches_sampled": per["FETCH_SIZE"][1], "seconds": round(time.time() - t0, 1),
                "source": "measured in THIS run: the device-resident leg of this command re-run under rocprofv3 --kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE "
                          "(separate passes, mean over the launches), counters -> bytes by the round-4 calibration"})
    return out


def c4_reverse_walk(w, subjects):
    """LookupResources(pod, view, user:U) for the C4 / C5 schema on the HOST, straight from the generator's arrays: the reverse walk the device runs
    (user -> the groups that hold it -> their ancestor groups -> what those view -> down the namespace arrow), in numpy without an index (masks over
    the edge arrays: a handful of subjects do not pay for a CSR over 100 M edges).  Three uses: every bit of the device's 8.45 M-pod result rows is
    compared with it, it prices the SURVEY 8(d) LookupResources byte model  17 + sum over reverse rows touched (8 + 4 deg) + N_pod / 8  (rows: the
    subject's five, three per visited group, one per visited namespace), and it is the tuned CPU figure beside the definition-based oracle.
    -> (list of sorted pod-id arrays, bytes per lookup, seconds per lookup)"""
    E = {(e[0], e[1], e[2]): (e[4], e[5]) for e in w.edges}
    gg_r, gg_s = E[("group", "member", "group")]
    gu_r, gu_s = E[("group", "member", "user")]
    pods, pod_ns = E[("pod", "namespace", "namespace")]
    pc_r, pc_s = E[("pod", "creator", "user")]
    nc_r, nc_s = E[("namespace", "creator", "user")]
    pvu_r, pvu_s = E[("pod", "viewer", "user")]
    pvg_r, pvg_s = E[("pod", "viewer", "group")]
    nvu_r, nvu_s = E[("namespace", "viewer", "user")]
    nvg_r, nvg_s = E[("namespace", "viewer", "group")]
    npod = w.nobjects["pod"]
    outs, nbytes, secs = [], [], []
    for u in subjects:
        t0 = time.perf_counter()
        u = np.uint32(u)
        edges = 0
        G = np.unique(gu_r[gu_s == u])
        edges += int(G.size)
        front = G
        while front.size:
            par = gg_r[np.isin(gg_s, front)]
            edges += int(par.size)
            par = np.unique(par)
            front = np.setdiff1d(par, G, assume_unique=True)
            G = np.union1d(G, front)
        p_direct = [pvu_r[pvu_s == u], pc_r[pc_s == u], pvg_r[np.isin(pvg_s, G)]]
        n_direct = [nvu_r[nvu_s == u], nc_r[nc_s == u], nvg_r[np.isin(nvg_s, G)]]
        edges += sum(int(a.size) for a in p_direct) + sum(int(a.size) for a in n_direct)
        NS = np.unique(np.concatenate(n_direct))
        via = pods[np.isin(pod_ns, NS)]
        edges += int(via.size)
        seen = np.zeros(npod, dtype=bool)
        for part in p_direct + [via]:
            seen[part] = True
        outs.append(np.flatnonzero(seen).astype(np.uint32))
        secs.append(time.perf_counter() - t0)
        nbytes.append(17 + 8 * (5 + 3 * int(G.size) + int(NS.size)) + 4 * edges + (npod + 7) // 8)
    return outs, np.asarray(nbytes, dtype=np.int64), np.asarray(secs)


def c5r_mixed_stream(args, w5, e5, gpu_perm, gpu_err, steps):
    """BASELINE configs[4]'s request stream on ONE replica: 90 % Check batches (262 144 host ids each, pinned buffers) / 10 %
    Filter requests -- LookupResources(pod, view, user:U), one subject per request as the reference issues it per LIST (pkg/authz/lookups.go:49-65),
    the 1 MB result row written into pinned host memory -- interleaved by the workload's seed (workloads.c5_stream) and issued by
    `--callers` threads as goroutines behind the shim would.  Every Check step's answers are compared with the device leg's (which the oracle checks
    item by item) and every lookup's row with the numpy reverse walk; k_rev_local's launch time comes from HIP events in a sequential pass."""
    import torch
    from aclgpu import workloads
    rt, perm_name, st = w5.check
    n = int(w5.res.size)
    ops = workloads.c5_stream(w5, steps)
    fsub = sorted({int(o_[1]) for o_ in ops if o_ != "C"})
    items0 = e5.make_items(rt, perm_name, w5.res, st, "", w5.subj)
    callers = max(1, args.callers)
    NB = callers + 1
    words = max(1, (e5.object_count(rt) + 31) // 32)
    hb = e5.host_alloc(NB * n * 21 + callers * (words * 4 + 8))
    h_items = hb[:NB * n * 16].view(aclgpu_item_dtype()).reshape(NB, n)
    h_perm = hb[NB * n * 16:NB * n * 17].reshape(NB, n)
    h_err = hb[NB * n * 17:NB * n * 21].view(np.int32).reshape(NB, n)
    lb0 = NB * n * 21
    lbufs = [(hb[lb0 + c * (words * 4 + 8):lb0 + c * (words * 4 + 8) + words * 4].view(np.uint32).reshape(1, words),
              hb[lb0 + c * (words * 4 + 8) + words * 4:lb0 + (c + 1) * (words * 4 + 8)].view(np.uint64)) for c in range(callers)]
    for b in range(NB):
        h_items[b] = np.roll(items0, b * 4099)
    # ---- sequential lookups with HIP events on: k_rev_local's launch time, and the rows every later step is compared with
    rows, cnts = {}, {}
    for s_ in fsub:  # warm (reverse rows are built on first use)
        e5.lookup_ids_batch(rt, perm_name, st, "", [s_], out=lbufs[0])
    e5.stats_reset()
    e5.set_timing(True)
    torch.cuda.synchronize()
    lat = []
    reps = max(1, 24 // max(1, len(fsub)))
    for _ in range(reps):
        for s_ in fsub:
            t1 = time.perf_counter()
            bm, ct = e5.lookup_ids_batch(rt, perm_name, st, "", [s_], out=lbufs[0])
            lat.append(time.perf_counter() - t1)
            rows[s_], cnts[s_] = bm[0].copy(), int(ct[0])
    e5.set_timing(False)
    stl = e5.stats()
    kname = "k_rev_local" if stl.get("rev_local_passes") else "k_rev_expand"
    kms, kn = (stl["rev_local_ms"], stl["rev_local_passes"]) if stl.get("rev_local_passes") else (stl["expand_ms"], stl["expand_launches"])
    walks, wbytes, wsecs = c4_reverse_walk(w5, fsub)
    npod = w5.nobjects[rt]
    bad_rows = 0
    for i, s_ in enumerate(fsub):
        got = np.flatnonzero(np.unpackbits(rows[s_].view(np.uint8), bitorder="little")[:npod]).astype(np.uint32)
        bad_rows += int(not np.array_equal(got, walks[i])) + int(cnts[s_] != walks[i].size)
    # ---- the stream, timed
    bad = []
    nxt = [0]
    lk = threading.Lock()
    go = [False]

    def run(ci):
        while not go[0]:
            time.sleep(0)
        while True:
            with lk:
                k = nxt[0]
                nxt[0] += 1
            if k >= len(ops):
                return
            if ops[k] == "C":
                b = ci  # (a caller's own batch buffers: NB > callers)
                e5.check_bulk_ids_into(h_items[b], h_perm[b], h_err[b])
                if not (np.array_equal(h_perm[b], np.roll(gpu_perm, b * 4099)) and np.array_equal(h_err[b], np.roll(gpu_err, b * 4099))):
                    bad.append(("C", k))
            else:
                s_ = int(ops[k][1])
                bm, ct = e5.lookup_ids_batch(rt, perm_name, st, "", [s_], out=lbufs[ci])
                if not (np.array_equal(bm[0], rows[s_]) and int(ct[0]) == cnts[s_]):
                    bad.append(("F", k, s_))

    for c in range(callers):  # every caller's context and buffers exist before the clock starts
        e5.check_bulk_ids_into(h_items[c], h_perm[c], h_err[c])
    # (the stream is 40 steps, 12-13 ms: one hiccup of the host -- a page fault storm, another thread's wake-up -- was a third of one run's figure, 404 M/s where three
    #  runs of the same binary on another box said 680-705.  Three passes, every answer of every pass compared; the MEDIAN pass is reported, all three recorded.)
    import gc
    passes = []
    for _rep in range(3):
        nxt[0] = 0
        go[0] = False
        ts = [threading.Thread(target=run, args=(c,)) for c in range(callers)]
        for t_ in ts:
            t_.start()
        torch.cuda.synchronize()
        gc_was = gc.isenabled()
        gc.disable()
        t0 = time.perf_counter()
        go[0] = True
        for t_ in ts:
            t_.join()
        torch.cuda.synchronize()
        passes.append(time.perf_counter() - t0)
        if gc_was:
            gc.enable()
    el = float(np.median(passes))
    e5.host_free(hb)
    nC = sum(1 for o_ in ops if o_ == "C")
    nF = len(ops) - nC
    tot_bytes = float(sum(wbytes[fsub.index(int(o_[1]))] for o_ in ops if o_ != "C"))
    per_launch = float(wbytes.mean())
    k_us = 1e3 * kms / max(1, kn)
    ach = per_launch / (k_us * 1e-6) / 1e9 if k_us > 0 else None
    return {"workload": "BASELINE configs[4] stream on one replica: 90 % Check batches (262 144 host ids) / 10 % Filter requests (LookupResources(pod, view, user:U) over 8.45 M pods, "
                        "one subject per request), interleaved by seed 0x5ACE0005; " + f"{callers} caller thread(s)",
            "steps": len(ops), "check_steps": nC, "filter_steps": nF, "seconds": round(el, 4), "passes_seconds": [round(x, 4) for x in passes], "ms_per_step": 1e3 * el / len(ops),
            "decisions_per_s": nC * n / el, "lookups_per_s": nF / el, "allowed_ids_per_lookup": float(np.mean([cnts[s_] for s_ in fsub])),
            "lookup_alone": {"p50_ms": 1e3 * float(np.median(lat)), "lookups_per_s": 1.0 / float(np.mean(lat)), "calls": len(lat), "result_row_bytes": int(words * 4),
                             "note": "one acl_lookup_resources_batch call of ONE subject at a time, pinned result row (the proxy's shape)"},
            "parity": {"check_steps_compared": 3 * nC, "lookups_compared": 3 * nF + len(fsub), "mismatches": len(bad) + bad_rows,
                       "checkers": "Check steps: the device leg's answers rotated (themselves compared with the oracle item by item); lookups: every bit of the result row against a "
                                   "numpy reverse walk over the generator's arrays, and against the oracle's DEFINITION on a pod sample (lookup_definition_sample)"},
            "roofline_lookup": {"bound": "hbm", "kernel": kname, "kernel_avg_us": k_us, "launches": int(kn), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": None, "algorithmic_bytes_per_lookup": per_launch,
                                "algorithmic_bytes_in_stream": tot_bytes, "result_rows": "HBM (the 8.45 M-pod type's row is 1 MB: beyond the block's LDS)",
                                "model": "SURVEY.md 8(d) LookupResources formula on the generator's arrays: 17 + sum over reverse rows touched (8 + 4 deg) + N_pod / 8"},
            "cpu_tuned_reverse_walk": {"value": 1.0 / float(wsecs.mean()), "unit": "lookups/s", "cores": 1,
                                       "sample": f"{len(fsub)} subjects, numpy masks over the generator's edge arrays (no index), one thread"},
            "_fsub": fsub, "_rows": rows}


def c5r_leg(args, local_rank, budget_s):
    """The 100 M-relationship replica (the HBM-regime data point: a 0.7 GB snapshot does not fit the 256 MiB Infinity Cache), in the DEFAULT run
    (VERDICT r4 next #2): device-resident kernel time over >= 10 launches (HIP events), every answer of the batch against the CPU oracle, the
    byte model's roofline and -- while the time budget lasts -- the counter traffic of the same command under rocprofv3 (two --pmc passes)."""
    import aclgpu
    from aclgpu import workloads
    t0 = time.time()
    w5 = workloads.c5(scale=1.0)
    t_gen = time.time() - t0
    e5 = aclgpu.Engine(w5.schema, device=local_rank, eager_contexts=True)
    w5.load(e5)
    e5.snapshot()
    t_load = time.time() - t0 - t_gen
    sub = argparse.Namespace(**vars(args))
    sub.workload, sub.scale, sub.batch, sub.replica = "C5", 1.0, 0, True
    r5, p5, er5 = check_bench(sub, w5, e5, max(10, min(args.steps, 40)), max(3, args.warmup), 1, 0, "C5R", "device")
    r5.pop("elapsed", None)
    r5.pop("value", None)
    snap_bytes = int(e5.stats()["snapshot_bytes"])
    mix = None
    if os.environ.get("ACL_BENCH_C5R_STREAM", "1") != "0":
        try:
            mix = c5r_mixed_stream(sub, w5, e5, p5, er5, max(40, min(args.steps, 80)))
        except Exception as ex:  # noqa: BLE001
            mix = {"error": f"{type(ex).__name__}: {ex}"}
    e5.close()
    r5["setup_s"] = {"generate": round(t_gen, 1), "load+snapshot": round(t_load, 1)}
    r5["snapshot_bytes"] = snap_bytes
    fsub, rows = (mix.pop("_fsub", []), mix.pop("_rows", {})) if mix else ([], {})

    def lookup_definition(o, cores):
        # LookupResources by its DEFINITION {pod : Check == HAS} on a fixed 50 000-pod sample per stream subject, by the oracle
        rt, perm_name, st = w5.check
        pods = np.unique(np.random.default_rng(55).integers(0, w5.nobjects[rt], size=50_000)).astype(np.uint32)
        bad = 0
        for s_ in fsub:
            lp, _le = o.check_bulk_ids_mt(cores, rt, perm_name, pods, st, "", np.full(pods.size, s_, dtype=np.uint32))
            bits = np.unpackbits(rows[s_].view(np.uint8), bitorder="little")
            bad += int(not np.array_equal(bits[pods] == 1, lp == 2))
        mix["parity"]["lookup_definition_sample"] = {"subjects": len(fsub), "pods_per_subject": int(pods.size), "mismatching_subjects": bad}
        mix["parity"]["mismatches"] += bad

    if not args.no_cpu:
        r5["_steps"] = max(10, min(args.steps, 40))
        sub.traffic = "measure" if (args.traffic == "measure" and time.time() - t0 < budget_s) else "static"
        cpu_and_roofline(sub, w5, r5, p5, er5, "C5R", with_oracle=lookup_definition if (mix and "parity" in mix) else None)
        r5.pop("_steps")
        r5["roofline"]["traffic_mode"] = sub.traffic
    else:
        r5.pop("kernel", None)
    if mix is not None:
        r5["mixed_stream"] = mix
        if "roofline_lookup" in mix:
            r5["lookups_per_s"] = mix["lookups_per_s"]
            r5["roofline_lookup"] = mix["roofline_lookup"]
            if "parity" in r5:
                r5["parity"]["mixed_stream_mismatches"] = mix["parity"]["mismatches"]
                r5["parity"]["mismatches"] += mix["parity"]["mismatches"]
    r5["seconds"] = round(time.time() - t0, 1)
    return r5


def launch_ranks(n, dry):
    """`python bench.py --gpus N` outside a launcher: start the N ranks here (python -m torch.distributed.run, one process per GPU,
    rendezvous on 127.0.0.1 -- the container's hostname may not resolve) and hand their exit code back.  Rank 0 prints the line."""
    import socket
    if not dry:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit(f"bench.py: {n} GPUs requested, {have} visible -- refusing to run fewer ranks and call it n_gpus={n}")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc:
        raise SystemExit(rc)


def dry_spawn(world, rank, local_rank):
    """--dry-spawn: the launch path without a GPU (tests/test_bench_spawn.py).  The ranks meet over gloo, agree on who is there, rank 0
    prints ONE line."""
    import torch
    import torch.distributed as dist
    import aclgpu
    from aclgpu import workloads
    seen = [rank]
    # what every rank of a real run does before and after its timed region, on a STORE-ONLY engine (no GPU): its own rotation of the request stream, the
    # graph loaded through the ABI, the type -> shard map of the 8-rank sharded leg, the max-over-ranks clock and the per-rank records -- over gloo
    w = workloads.c1()
    shift = (rank * 32749) % max(1, int(w.res.size))
    res = np.roll(w.res, shift)
    eng = aclgpu.Engine(w.schema, store_only=True)
    w.load(eng)
    owners = {t_: int(eng._L.acl_shard_of_type(eng._h, eng.type_id(t_))) for t_ in ("user", "namespace")}
    if world > 1:
        eng._check(eng._L.acl_shard_configure(eng._h, rank, world))
        owners = {t_: int(eng._L.acl_shard_of_type(eng._h, eng.type_id(t_))) for t_ in ("user", "namespace")}
    eng.close()
    elapsed = 1.0 + 0.01 * rank
    per_rank = [{"rank": rank, "elapsed_s": elapsed, "first_resource": int(res[0])}]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = 1 + local_rank
        dist.all_reduce(t)
        seen = [int(x) - 1 for x in t]
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)  # the contract's max-over-ranks timing
        elapsed = float(tt.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "elapsed_s": 1.0 + 0.01 * rank, "first_resource": int(res[0])})
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_spawn": True, "n_gpus": world, "local_ranks_seen": seen, "metric": "check_decisions_per_sec", "value": None, "scaling": "weak",
                          "max_elapsed_s": elapsed, "per_rank": per_rank, "shard_of_type": owners,
                          "sharded_leg": "on" if world == 8 else "off (auto: the type-hash sharded leg runs at 8 ranks; replicas are `value` at every N)",
                          "distinct_request_streams": len({r_["first_resource"] for r_ in per_rank})}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (one process per GPU).  Under torchrun it must equal WORLD_SIZE; "
                    "without torchrun and N > 1 this process launches the N ranks itself (torch.distributed.run, 127.0.0.1)")
    ap.add_argument("--dry-spawn", action="store_true", help="launch plumbing only (no GPU): the N ranks rendezvous over gloo and rank 0 prints one line")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="C4", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU-oracle sample time (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--strings", default="on", choices=["on", "off"], help="name every pod and user (string-path leg on named objects); off: ids only")
    ap.add_argument("--numa", default="gpu", choices=["gpu", "off"], help="gpu: every rank runs on the NUMA node of its GPU (pinned request / answer arrays and name tables are then local to it)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "measure", "static"],
                    help="roofline.traffic: measure = two rocprofv3 --pmc passes of this command's device leg, now (about 25 s); static = profiles/traffic.json "
                         "(an earlier run's passes, labelled); auto = measure for the headline workload at N=1 with the CPU legs on")
    ap.add_argument("--replica", action="store_true", help="with --workload C5: the 100 M-relationship graph as one unsharded replica (the beyond-L3 data point)")
    ap.add_argument("--stream", action="store_true", help="with --workload C5 --replica: also run the 90 / 10 Check + Filter stream (what the default run's C5R leg does)")
    ap.add_argument("--legs", default="all", choices=["all", "device"], help="device: only the device-resident leg (for rocprofv3 runs: every k_expand "
                    "launch of the process is then a sequential one, so the profiler's average equals the roofline's)")
    ap.add_argument("--native-loop", default="on", choices=["on", "off"], help="sharded leg: also time the level loop inside libaclgpu.so (acl_shard_check_bulk)")
    ap.add_argument("--pipeline", default="blocking", choices=["blocking", "submit"], help="how the timed host-id leg keeps batches in flight")
    ap.add_argument("--callers", type=int, default=3, help="--pipeline blocking: host threads issuing blocking acl_check_bulk_ids calls (default 3 = the batches the engine admits at once: its chain lanes; Python threads, so a native caller needs fewer)")
    ap.add_argument("--devices", default="", help="ONE process in front of several GPUs: comma-separated HIP ordinals of the engine's in-process replicas "
                    "(acl_open_replicas: one relationship store, one HBM snapshot per entry; an ordinal may repeat: logical replicas on one GPU).  A step is then one "
                    "batch PER replica, `--callers` is per replica.")
    ap.add_argument("--one-process", action="store_true", help="with --gpus N: do not launch N ranks; one process with the replicas 0..N-1 (= --devices 0,...,N-1)")
    ap.add_argument("--window", type=int, default=2, help="--pipeline submit: batches in flight (<= the engine's evaluation contexts)")
    ap.add_argument("--configs", default="auto", choices=["auto", "on", "off"], help="also measure C2 and C3 in the same run (auto: at N=1 with the default workload)")
    ap.add_argument("--sharded", default="auto", choices=["auto", "on", "off"],
                    help="extra leg: the SAME graph partitioned by type hash over the ranks, per-level RCCL all-gather of cross-shard "
                         "frontiers (north star's 8-GPU layout).  auto = on at 8 ranks.  Reported beside `value`, never as `value`.")
    ap.add_argument("--sharded-isolate", default="auto", choices=["auto", "on", "off"], help="run the sharded leg in child processes (auto: when there is more than one rank)")
    ap.add_argument("--sharded-child", default="", help=argparse.SUPPRESS)  # internal: this process IS such a child; rank 0 writes its result there
    ap.add_argument("--logical-shards", type=int, default=0, help="with 1 GPU: run the sharded leg as G logical shards on this device (emulated)")
    ap.add_argument("--exchange", default="both", choices=["allgather", "alltoall", "both"],
                    help="sharded leg: how Check frontiers cross shards (allgather = the north star's form; alltoall moves G x fewer bytes)")
    args = ap.parse_args()

    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    args.replica_devices = [int(x) for x in args.devices.split(",") if x.strip() != ""]
    if args.one_process and not args.replica_devices:
        args.replica_devices = list(range(args.gpus or 1))
    if args.replica_devices:
        if under_launcher:
            raise SystemExit("bench.py: --devices / --one-process is the ONE-process mode; do not start it under a launcher")
        args.gpus = None  # (n_gpus of the line = distinct devices of the replica set)
    if not under_launcher and (args.gpus or 1) > 1:
        return launch_ranks(args.gpus, args.dry_spawn)  # this process is the launcher: the ranks print, it forwards their exit code

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a line whose n_gpus is not what was asked for")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_spawn:
        return dry_spawn(world, rank, local_rank)
    if args.traffic == "auto":  # (every rank of an N > 1 run would need a profiler pass of its own; the other configurations keep the committed figures)
        args.traffic = "measure" if (world == 1 and not args.no_cpu and args.legs == "all" and not args.sharded_child) else "static"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU evaluation path")
    if torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} GPUs requested, {torch.cuda.device_count()} visible (one process per GPU: rank {rank} has no device)")
    torch.cuda.set_device(local_rank)
    numa_node = bind_to_gpu_numa_node(local_rank) if args.numa == "gpu" else None
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
    if world > 1 and rank:
        # every rank answers its own request stream against its replica: the SAME request mix in a different order (the (resource,
        # subject) pairs rotated together -- rotating one column against the other would turn every engineered hit into a random
        # pair, a different workload).  Same multiset of requests on every rank => rank 0's algorithmic bytes are every rank's.
        shift = (rank * 32749) % max(1, int(w.res.size))
        w.res, w.subj = np.roll(w.res, shift), np.roll(w.subj, shift)
    t_gen = time.time() - t0
    n = int(w.res.size)

    if args.sharded_child:  # the isolated sharded leg (isolated_sharded_leg): nothing else, no stdout
        global SHARDED_CHECKPOINT
        res = {}

        def save(r):
            tmp = args.sharded_child + ".tmp"
            with open(tmp, "w") as f:
                json.dump(r, f)
            os.replace(tmp, args.sharded_child)

        if rank == 0:
            SHARDED_CHECKPOINT = save
        eng_r = aclgpu.Engine(w.schema, device=local_rank)
        w.load(eng_r)
        try:
            sharded_leg(args, w, eng_r, canon_res, canon_subj, world, rank, local_rank, res)
        except Exception as ex:  # noqa: BLE001
            res["error"] = f"{type(ex).__name__}: {ex}"
        finally:
            eng_r.close()
        if rank == 0:
            save(res)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload == "C5" and not args.replica:
        return c5_bench(args, w, world, rank, local_rank, t_gen)
    label = "C5R" if args.workload == "C5" else args.workload
    # (every evaluation context exists before any clock starts: one made on demand inside the 20-step timed region -- the first time three callers
    #  really overlap -- costs it 3 ms of allocations under the pool's lock: 0.38 instead of 0.20 ms per step in one of this round's runs)
    eng = aclgpu.Engine(w.schema, device=local_rank, contexts=max(2, args.window, args.callers) + 1, devices=args.replica_devices or None, eager_contexts=True)
    t0 = time.time()
    if args.legs == "all" and rank == 0 and args.workload != "C3" and args.strings != "off":
        name_objects(eng, w)  # (before the load: ids follow interning order)
    t_names = time.time() - t0
    t0 = time.time()
    w.load(eng)
    eng.snapshot()
    t_load = time.time() - t0
    if args.workload == "C3":
        out = filter_bench(args, w, eng, args.steps, args.warmup)
        out.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                    "config": {"workload": out.pop("workload"), "lookups_per_step": out.pop("lookups_per_step"), "relationships": out.pop("relationships"),
                               "objects": out.pop("objects"), "scale": args.scale}})
        if rank == 0:
            print(json.dumps(out))
        eng.close()
        if out.get("parity", {}).get("mismatches"):
            raise SystemExit("PARITY FAILURE: GPU lookup differs from the oracle")
        return

    rec, gpu_perm, gpu_err = check_bench(args, w, eng, args.steps, args.warmup, world, rank, label, args.legs, dist if world > 1 else None)
    elapsed = rec.pop("elapsed")
    if label == "C5R" and args.stream and rank == 0:
        mm = c5r_mixed_stream(args, w, eng, gpu_perm, gpu_err, max(40, args.steps))
        mm.pop("_fsub", None)
        mm.pop("_rows", None)
        rec["mixed_stream"] = mm
    per_rank = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per_rank = [None] * world  # every rank's own clock and kernel time: the roofline is per GPU
        k_ = rec.get("kernel") or {}
        dist.all_gather_object(per_rank, {"rank": rank, "elapsed_s": elapsed, "decisions_per_s": n * args.steps / elapsed,
                                          "kernel": k_.get("name"), "kernel_avg_us": 1e3 * k_["ms"] / k_["launches"] if k_.get("launches") else None,
                                          "device_resident_decisions_per_s": rec["device_resident"]["decisions_per_s"]})
        elapsed = float(tt.item())
    stats = eng.stats()

    # ---- extra leg (outside the timed region above): the sharded graph
    want_sharded = (args.sharded == "on" or (args.sharded == "auto" and world == 8)) and (world > 1 or args.logical_shards > 1)

    out = None
    if rank == 0:
        nrep = max(1, len(args.replica_devices))  # one process, in-process replicas (--devices): a step is one batch per replica
        total = n * args.steps * world * nrep
        rec.pop("value")
        out = {
            "metric": "check_decisions_per_sec", "value": total / elapsed, "unit": "decisions/s", "n_gpus": len(set(args.replica_devices)) if args.replica_devices else world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": rec.pop("workload"), "batch_per_gpu": rec.pop("batch"), "relationships": rec.pop("relationships"),
                       "objects": rec.pop("objects"), "scale": args.scale, "parallelism": (f"ONE process, {nrep} in-process replica(s) on device(s) {args.replica_devices}: one relationship store, one HBM snapshot per replica (acl_open_replicas)"
                                                                      if args.replica_devices else f"replicas x{world} (request-level data parallel)"), "numa_node_of_rank0": numa_node,
                       "timed": "host-id ABI calls (host buffers in, host buffers out: PCIe both ways inside the call) over 8 distinct pinned batches, " + (f"submit/wait window {args.window}" if args.pipeline == "submit" else f"{args.callers * nrep} blocking caller thread(s)") if args.legs == "all"
                                else "device-resident calls only (--legs device)"},
            "p50_batch_ms": rec.get("latency", {}).get("p50_batch_ms", rec["device_resident"]["p50_batch_ms"]),
            "setup_s": {"generate": round(t_gen, 2), "name_objects": round(t_names, 2), "load+snapshot": round(t_load, 2)},
            "snapshot_bytes": int(stats["snapshot_bytes"]),
        }
        kernel = rec.get("kernel")
        out.update(rec)
        if not args.no_cpu:
            # at EVERY N: rank 0 times the CPU oracle on its own batch and prices the roofline from the oracle's byte model (the other ranks
            # wait at the barrier below -- outside every timed region)
            out["_steps"] = args.steps
            cpu_and_roofline(args, w, out, gpu_perm, gpu_err, label)
            out.pop("_steps")
            tk, lr = out.get("timed_leg_kernels") or {}, (out.get("host_ids") or {}).get("long_run") or {}
            # the timed leg reconciled with the roofline's sequential kernel time, and the >= 200-step figure, as flat numbers inside `roofline`
            out["roofline"].update({"kernel_us_in_leg": tk.get("kernel_us_in_leg"), "overlap_factor": tk.get("overlap_factor"),
                                    "long_run_decisions_per_s": lr.get("decisions_per_s"), "long_run_ms_per_step": lr.get("ms_per_step"), "long_run_steps": lr.get("steps")})
            out["long_run"] = lr or None
            if per_rank:
                rf = out["roofline"]
                bpl = rf["algorithmic_bytes_per_launch"]
                gl = []
                for pr_ in per_rank:  # same request multiset on every rank (rotated): rank 0's bytes per launch are every rank's
                    a_ = bpl / (pr_["kernel_avg_us"] * 1e-6) / 1e9 if pr_.get("kernel_avg_us") else None
                    gl.append({"rank": pr_["rank"], "kernel": pr_["kernel"], "kernel_avg_us": pr_["kernel_avg_us"], "achieved": a_,
                               "frac": a_ / HBM_PEAK_GBS if a_ else None, "host_id_decisions_per_s": pr_["decisions_per_s"],
                               "device_resident_decisions_per_s": pr_["device_resident_decisions_per_s"]})
                ok_ = [g_["achieved"] for g_ in gl if g_["achieved"]]
                rf["per_gpu"] = gl
                rf["achieved"] = float(np.mean(ok_)) if ok_ else None  # per GPU (mean over ranks) against ONE GPU's peak
                rf["frac"] = rf["achieved"] / HBM_PEAK_GBS if ok_ else None
                rf["frac_min_over_gpus"] = min(ok_) / HBM_PEAK_GBS if ok_ else None
                rf["aggregate_achieved"] = float(np.sum(ok_)) if ok_ else None
                rf["note"] = "achieved / frac are PER GPU (mean over the ranks' own HIP-event kernel times); every rank runs the same request multiset"
        else:
            out.pop("kernel", None)
            rot = out.pop("rotations", None)
            if rot:
                out["parity"] = {"checked_against_oracle": 0, "distinct_batches_checked": rot["batches"], "mismatches_vs_rotated_batch0": rot["mismatches_vs_rotated_batch0"]}
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                               "kernel": kernel["name"] if kernel else None, "kernel_avg_us": 1e3 * kernel["ms"] / kernel["launches"] if kernel else None,
                               "note": "--no-cpu: algorithmic bytes need the CPU oracle's byte model"}
            out["cpu_baseline"] = None
    eng.close()

    # ---- the other single-GPU BASELINE configurations, same run (never part of `value`)
    if rank == 0 and world == 1 and args.legs == "all" and (args.configs == "on" or (args.configs == "auto" and args.workload == "C4" and args.scale == 1.0 and not args.batch)):
        cfgs = {}
        try:
            w2 = workloads.c2()
            e2 = aclgpu.Engine(w2.schema, device=local_rank, contexts=max(2, args.window, args.callers) + 1, eager_contexts=True)
            if args.strings != "off":
                name_objects(e2, w2)
            w2.load(e2)
            e2.snapshot()
            r2, p2, er2 = check_bench(args, w2, e2, max(args.steps, 50), args.warmup, 1, 0, "C2", "all")
            r2.pop("elapsed")
            r2["metric"], r2["unit"] = "check_decisions_per_sec", "decisions/s"
            if not args.no_cpu:
                r2["_steps"] = max(args.steps, 50)
                tmode, args.traffic = args.traffic, "static"
                cpu_and_roofline(args, w2, r2, p2, er2, "C2")
                args.traffic = tmode
                r2.pop("_steps")
            else:
                r2.pop("kernel", None)
            e2.close()
            cfgs["C2"] = r2
            w3 = workloads.c3()
            e3 = aclgpu.Engine(w3.schema, device=local_rank)
            w3.load(e3)
            e3.snapshot()
            cfgs["C3"] = filter_bench(args, w3, e3, max(args.steps, 20), args.warmup)
            e3.close()
            if args.strings != "off":
                cfgs["C3"]["postfilter_one_user"] = postfilter_c3_leg(local_rank)
        except Exception as ex:  # noqa: BLE001 -- the headline line is printed whatever happens here
            cfgs["error"] = f"{type(ex).__name__}: {ex}"
        # the 100 M-relationship replica: the honest HBM-regime point beside the cache-resident headline (its own time budget: ~1-2 minutes)
        if os.environ.get("ACL_BENCH_C5R", "1") != "0":
            try:
                cfgs["C5R"] = c5r_leg(args, local_rank, float(os.environ.get("ACL_BENCH_C5R_BUDGET_S", "150")))
            except Exception as ex:  # noqa: BLE001
                cfgs["C5R"] = {"error": f"{type(ex).__name__}: {ex}"}
        out["configs"] = cfgs
        out["single_checks"] = single_checks_leg()
        rf5 = (cfgs.get("C5R") or {}).get("roofline") or {}
        if out.get("roofline") is not None and rf5:  # flat, inside `roofline`: what the driver's record keeps of this line
            out["roofline"].update({"c5r_kernel_avg_us": rf5.get("kernel_avg_us"), "c5r_frac": rf5.get("frac"), "c5r_achieved": rf5.get("achieved"), "c5r_traffic": rf5.get("traffic"),
                                    "c5r_traffic_frac": rf5.get("traffic_frac"), "c5r_parity_checked": (cfgs["C5R"].get("parity") or {}).get("checked_against_oracle"),
                                    "c5r_parity_mismatches": (cfgs["C5R"].get("parity") or {}).get("mismatches")})
            rl5 = cfgs["C5R"].get("roofline_lookup") or {}
            ms5 = cfgs["C5R"].get("mixed_stream") or {}
            if rl5:  # the 90 / 10 Check + Filter stream on the replica: the reverse kernel's own roofline entry and the stream's rates
                out["roofline"].update({"c5r_stream_decisions_per_s": ms5.get("decisions_per_s"), "c5r_stream_lookups_per_s": ms5.get("lookups_per_s"),
                                        "c5r_lookup_kernel": rl5.get("kernel"), "c5r_lookup_kernel_avg_us": rl5.get("kernel_avg_us"), "c5r_lookup_frac": rl5.get("frac"),
                                        "c5r_lookup_achieved": rl5.get("achieved"), "c5r_stream_mismatches": (ms5.get("parity") or {}).get("mismatches")})

    # ---- extra leg (outside the timed region, after the main line is complete): the sharded graph.  Whatever happens in
    # it -- an exception on this rank, a wedged collective -- the main line is still printed exactly once.
    if want_sharded and (args.sharded_isolate == "on" or (args.sharded_isolate == "auto" and world > 1)):
        sh = isolated_sharded_leg(rank, world)
        if rank == 0:
            out["sharded"] = sh
            print(json.dumps(out), flush=True)
    elif want_sharded:
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
        eng_r = aclgpu.Engine(w.schema, device=local_rank)
        w.load(eng_r)
        try:
            emit(sharded_leg(args, w, eng_r, canon_res, canon_subj, world, rank, local_rank, sharded_result))
        except Exception as ex:  # noqa: BLE001
            sharded_result["error"] = f"{type(ex).__name__}: {ex}"
            emit(sharded_result)
        finally:
            timer.cancel()
            eng_r.close()
    elif rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    bad = out and (out.get("parity", {}).get("mismatches") or any(isinstance(c, dict) and c.get("parity", {}).get("mismatches") for c in out.get("configs", {}).values()))
    if bad:
        raise SystemExit("PARITY FAILURE: GPU answers differ from the oracle")


if __name__ == "__main__":
    main()
