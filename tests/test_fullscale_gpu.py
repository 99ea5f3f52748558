"""Full-size parity (run with `-m gpu` on an MI355X): BASELINE.json's configurations at scale 1.0, EVERY answer compared.

VERDICT r1 "what's weak" #1: the GPU tests stopped at C2 x0.05 / C3 x0.05 / C4 x0.02 and full-size parity lived only in
bench.py side-legs with small samples.  Here:
  C2  1 M relationships, all 65 536 answers            vs the CPU oracle (multi-threaded bulk check)
  C3  all 64 power users' LookupResources bitmaps      vs the oracle's DEFINITION {id : Check == HAS} over all 98 990 pods
      (6.3 M oracle checks, multi-threaded)
  C4  10 M relationships, all 262 144 answers (perm AND err), through the host-id ABI call, the device-resident call,
      the pipelined submit/wait path and the string path (a 16 384-item slice: it interns 5 strings per item)
  C5  100 M relationships over 8 logical shards on one device (emulated layout), all 262 144 answers of a Check batch
      + one Filter bitmap vs the oracle (ACL_SKIP_C5_FULL=1 skips it: it needs ~25 GB of host memory and a minute)
The oracle is the checker here, never the thing under test (oracle/ header)."""
import os

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


def host_threads():
    c = max(1, len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            c = max(1, min(c, int(int(q) / int(per) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return min(c, 32)


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


def load_both(aclgpu, w):
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    e = aclgpu.Engine(w.schema)
    w.load(e)
    return o, e


def test_c2_full(aclgpu):
    from aclgpu import workloads
    w = workloads.c2()
    assert 990_000 <= w.ntuples <= 1_010_000 and w.res.size == 65536
    o, e = load_both(aclgpu, w)
    with e:
        rt, perm, st = w.check
        p, er = e.check_bulk_ids(e.make_items(rt, perm, w.res, st, "", w.subj))
        op, oe = o.check_bulk_ids_mt(host_threads(), rt, perm, w.res, st, "", w.subj)
        assert np.array_equal(p, op) and np.array_equal(er, oe)
        assert 0.3 < (p == 2).mean() < 0.9


def test_c3_full_all_power_users(aclgpu):
    from aclgpu import workloads
    w = workloads.c3()
    assert w.lookup_subjects.size == 64
    o, e = load_both(aclgpu, w)
    with e:
        rt, perm, st = w.check
        bms, counts = e.lookup_ids_batch(rt, perm, st, "", w.lookup_subjects)
        npod = w.nobjects[rt]
        pods = np.arange(npod, dtype=np.uint32)
        nt = host_threads()
        for i, s in enumerate(w.lookup_subjects):
            op, oe = o.check_bulk_ids_mt(nt, rt, perm, pods, st, "", np.full(npod, s, dtype=np.uint32))
            want = np.flatnonzero(op == 2).astype(np.uint32)  # the definition of LookupResources
            got = np.flatnonzero(np.unpackbits(bms[i].view(np.uint8), bitorder="little")).astype(np.uint32)
            assert np.array_equal(got, want), (i, int(s), got.size, want.size)
            assert counts[i] == want.size
            assert not oe.any()
        assert 5_000 < counts.mean() < 20_000  # "~10 k allowed ids per user"
        # the same 64 lookups one by one (the proxy's shape) give the same bitmaps
        for i in (0, 31, 63):
            one, _ = e.lookup_ids_batch(rt, perm, st, "", [int(w.lookup_subjects[i])])
            assert np.array_equal(one[0], bms[i])


def test_c4_full_every_entry_point(aclgpu):
    import torch
    from aclgpu import workloads
    w = workloads.c4()
    assert 9_900_000 <= w.ntuples <= 10_100_000 and w.res.size == 262144 and sum(w.nobjects.values()) == 1_000_000
    o, e = load_both(aclgpu, w)
    with e:
        rt, perm, st = w.check
        n = w.res.size
        op, oe = o.check_bulk_ids_mt(host_threads(), rt, perm, w.res, st, "", w.subj)
        items = e.make_items(rt, perm, w.res, st, "", w.subj)
        # (ii) host-id ABI call
        p, er = e.check_bulk_ids(items)
        assert np.array_equal(p, op) and np.array_equal(er, oe)
        # (i) device-resident call
        d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
        d_perm = torch.zeros(n, dtype=torch.uint8, device="cuda")
        d_err = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        e.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
        assert np.array_equal(d_perm.cpu().numpy(), op) and np.array_equal(d_err.cpu().numpy(), oe)
        # pipelined: 8 distinct batches (rotations of the request stream) in flight over the engine's contexts, pinned buffers
        k = 8
        hb = e.host_alloc(k * n * 21)
        h_items = hb[:k * n * 16].view(aclgpu.ITEM_DTYPE).reshape(k, n)
        h_perm = hb[k * n * 16:k * n * 17].reshape(k, n)
        h_err = hb[k * n * 17:].view(np.int32).reshape(k, n)
        for b in range(k):
            h_items[b] = np.roll(items, b * 4099)
        tickets = [e.submit_ids(h_items[b], h_perm[b], h_err[b]) for b in range(k)]
        for t in tickets:
            e.wait(t)
        for b in range(k):
            assert np.array_equal(h_perm[b], np.roll(op, b * 4099)) and np.array_equal(h_err[b], np.roll(oe, b * 4099)), b
        # chip-filling batches from several threads at once (goroutines behind the cgo shim): their kernels are chained on the
        # device, each stream waiting for the event behind the previous batch's kernel; four callers exercise the in-flight cap
        import threading
        for b in range(k):
            h_perm[b][:] = 255
            h_err[b][:] = -1
        def caller(mine):
            for b in mine:
                e.check_bulk_ids_into(h_items[b], h_perm[b], h_err[b])
        ths = [threading.Thread(target=caller, args=(range(t, k, 4),)) for t in range(4)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for b in range(k):
            assert np.array_equal(h_perm[b], np.roll(op, b * 4099)) and np.array_equal(h_err[b], np.roll(oe, b * 4099)), b
        e.host_free(hb)
        # (iii) string path (anonymous numeric ids have no names: name a slice of the objects first)
        m = 16384
        # bulk-loaded ids are anonymous; the string path needs names -> a small named graph is covered by the KATs and
        # test_workload_parity; here the interning path is exercised at size with unknown names (all NO_PERMISSION) ...
        strs = [("pod", f"nope-{i}", "view", "user", f"nobody-{i}", "") for i in range(m)]
        sp, se_ = e.check_bulk(strs)
        assert set(sp) == {1} and not any(se_)
        # small batches take the single-launch kernel: same answers as the big batch's prefix, at every size around a wave
        e.stats_reset()
        for sz in (1, 2, 63, 64, 65, 127, 1000, 4096, 8192):
            ps, es = e.check_bulk_ids(items[:sz])
            assert np.array_equal(ps, op[:sz]) and np.array_equal(es, oe[:sz]), sz
        assert e.stats()["local_passes"] >= 1


def test_c4_full_named_objects_through_strings(aclgpu):
    """The string entry points at FULL size on named objects ("the full-size string path is only exercised with unknown names"):
    every pod and user of the 10 M-relationship graph gets a name (ids follow interning order, so the bulk-loaded relationships are theirs), and the
    whole 262 144-item batch goes through acl_check_bulk_v ({pointer, length} fields) and a 65 536-item slice through acl_check_bulk (C strings):
    every answer equals the oracle's answer for the ids those names stand for."""
    import ctypes
    from aclgpu import workloads
    w = workloads.c4()
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rt, perm, st = w.check
    op, oe = o.check_bulk_ids_mt(host_threads(), rt, perm, w.res, st, "", w.subj)
    pod_ns = np.zeros(w.nobjects["pod"], dtype=np.int64)
    for e_ in w.edges:
        if e_[0] == "pod" and e_[1] == "namespace":
            pod_ns[e_[4]] = e_[5]
    names = {"pod": [f"ns{int(pod_ns[i])}/pod-{i}" for i in range(w.nobjects["pod"])], "user": [f"user-{i}" for i in range(w.nobjects["user"])]}
    with aclgpu.Engine(w.schema) as e:
        out = ctypes.c_uint32()
        for t_, ns_ in names.items():
            tid = e.type_id(t_)
            for nm in ns_:
                e._check(e._L.acl_intern(e._h, tid, nm.encode(), ctypes.byref(out)))
            assert out.value == len(ns_) - 1
        w.load(e)
        qs = [("pod", names["pod"][int(r)], "view", "user", names["user"][int(s_)], "") for r, s_ in zip(w.res, w.subj)]
        pv, ev = e.check_bulk_views(e.make_check_views(qs))
        assert np.array_equal(pv, op) and np.array_equal(ev, oe)
        m = 65536
        pc, ec = e.check_bulk_prepared(e.make_check_strings_named(qs[:m]))
        assert np.array_equal(pc, op[:m]) and np.array_equal(ec, oe[:m])
        assert 0.5 < (pv == 2).mean() < 0.95
        # LookupResources by name on the same graph: the names of the result row are the names of the ids the numeric lookup returns (which
        # test_lookup_local_gpu / test_c3_full check against the oracle; its brute-force definition over 845 000 pods would take a minute here)
        u = int(w.subj[0])
        got = e.lookup("pod", "view", "user", names["user"][u])
        ids = e.lookup_ids("pod", "view", "user", "", u)
        assert got == {names["pod"][int(i)] for i in ids} and len(got) == ids.size > 0
        assert all(op[k] == 2 for k in np.flatnonzero(w.subj == u) if int(w.res[k]) in set(ids.tolist())) and any(int(w.res[k]) in set(ids.tolist()) for k in np.flatnonzero(w.subj == u))


C5_SKIP = pytest.mark.skipif(os.environ.get("ACL_SKIP_C5_FULL") == "1", reason="ACL_SKIP_C5_FULL=1")


@pytest.fixture(scope="module")
def c5(aclgpu):
    """BASELINE config 5's graph (100 M relationships / 10 M objects) with the oracle's answers, built ONCE for the tests below: the Check
    batch's 262 144 answers, and for eight LookupResources subjects -- deep-group members (the stream's own lookup subjects), a direct
    viewer of namespaces, a pod creator, and a user nobody has a relationship with -- the DEFINITION {pod : Check == HAS} over a fixed
    sample of 400 000 pods (all 8.45 M pods x 8 subjects would keep the oracle busy for minutes).  The oracle stays loaded for the one-replica
    test (it also confirms ids the engine returns OUTSIDE the sample) and is dropped before the 8-shard test needs the memory."""
    from aclgpu import workloads
    w = workloads.c5()
    assert 99_000_000 <= w.ntuples <= 101_000_000
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rt, perm, st = w.check
    nt = host_threads()
    op, oe = o.check_bulk_ids_mt(nt, rt, perm, w.res, st, "", w.subj)
    E = {(e[0], e[1], e[2]): (e[4], e[5]) for e in w.edges}
    ns_viewer = int(np.bincount(E[("namespace", "viewer", "user")][1]).argmax())  # the user that views the most namespaces directly
    creator = int(E[("pod", "creator", "user")][1][12345])
    nobody = int(w.nobjects["user"]) + 7  # no relationship names this id
    subs = np.asarray([int(x) for x in w.lookup_subjects[:5]] + [ns_viewer, creator, nobody], dtype=np.uint32)
    rng = np.random.default_rng(5)
    pods = np.unique(rng.integers(0, w.nobjects[rt], size=400_000)).astype(np.uint32)
    lp = [o.check_bulk_ids_mt(nt, rt, perm, pods, st, "", np.full(pods.size, s, dtype=np.uint32))[0] for s in subs]
    d = {"w": w, "o": o, "op": op, "oe": oe, "subs": subs, "pods": pods, "lp": lp}
    yield d
    d.clear()


@C5_SKIP
def test_c5_full_one_replica_mixed_stream(aclgpu, c5):
    """BASELINE configs[4]'s workload on ONE replica: the 100 M-relationship graph unsharded -- what every GPU of the
    replica layout holds -- answering the mixed stream: 256 k-item Check batches and LookupResources(pod, view, user:U) over the
    8.45 M-pod type, whose result rows (1 MB each) do not fit the block's LDS, so k_rev_local keeps them in HBM (`lds_words == 0`: the path
    tests/test_lookup_local_gpu.py only reaches at scale 0.02 behind ACL_REV_LDS_ROWS=0).  The reference issues exactly this call per LIST
    (pkg/authz/lookups.go:49-65).  Checked: every Check answer; per lookup the DEFINITION on the 400 000-pod sample, the id count against
    the bitmap, and up to 50 000 of the returned ids OUTSIDE the sample against the oracle (no false grants anywhere)."""
    from aclgpu import workloads
    w, o, subs, pods = c5["w"], c5["o"], c5["subs"], c5["pods"]
    rt, perm, st = w.check
    nt = host_threads()
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        items = e.make_items(rt, perm, w.res, st, "", w.subj)
        p, er = e.check_bulk_ids(items)
        assert np.array_equal(p, c5["op"]) and np.array_equal(er, c5["oe"])
        e.stats_reset()
        bms, counts = e.lookup_ids_batch(rt, perm, st, "", subs)
        stl = e.stats()
        assert stl["rev_local_passes"] >= 1, stl  # the single-launch reverse walk took the batch (result rows in HBM)
        in_sample = np.zeros(w.nobjects[rt], dtype=bool)
        in_sample[pods] = True
        for i, s in enumerate(subs):
            bits = np.unpackbits(bms[i].view(np.uint8), bitorder="little")[:w.nobjects[rt]]
            assert np.array_equal(bits[pods] == 1, c5["lp"][i] == 2), (i, int(s))
            assert int(counts[i]) == int(bits.sum()), (i, int(s))
            extra = np.flatnonzero((bits == 1) & ~in_sample).astype(np.uint32)[:50_000]
            if extra.size:
                xp, _xe = o.check_bulk_ids_mt(nt, rt, perm, extra, st, "", np.full(extra.size, s, dtype=np.uint32))
                assert (xp == 2).all(), (i, int(s), int((xp != 2).sum()))
        assert int(counts[-1]) == 0 and int(counts[:5].min()) > 1000, counts  # nobody sees nothing; a deep-group member sees thousands of pods
        # one lookup per call (the proxy's shape) returns the batch's row
        for i in (0, 5, 7):
            one, c1 = e.lookup_ids_batch(rt, perm, st, "", [int(subs[i])])
            assert np.array_equal(one[0], bms[i]) and c1[0] == counts[i]
        # the mixed stream itself (90 % Check batches / 10 % Filter requests, interleaved by the workload's seed), two callers
        # at once as goroutines behind the shim would be: every step's answer equals the sequential one
        import threading
        ops = workloads.c5_stream(w, 24)
        assert any(x == "C" for x in ops) and any(x != "C" for x in ops)
        bad = []

        def caller(mine):
            for k in mine:
                if ops[k] == "C":
                    pk, ek = e.check_bulk_ids(np.roll(items, k * 4099))
                    if not (np.array_equal(pk, np.roll(c5["op"], k * 4099)) and np.array_equal(ek, np.roll(c5["oe"], k * 4099))):
                        bad.append(("C", k))
                else:
                    j = k % subs.size
                    bk, ck = e.lookup_ids_batch(rt, perm, st, "", [int(subs[j])])
                    if not (np.array_equal(bk[0], bms[j]) and ck[0] == counts[j]):
                        bad.append(("F", k, j))

        ths = [threading.Thread(target=caller, args=(range(t, len(ops), 2),)) for t in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not bad, bad


@C5_SKIP
def test_c5_full_emulated_8_shards(aclgpu, c5):
    """100 M relationships / 10 M objects, hash(type) mod 8 on ONE device (emulated layout): all 262 144
    answers of a Check batch and one Filter bitmap against the oracle."""
    from aclgpu import sharded
    w, op, oe, pods = c5["w"], c5["op"], c5["oe"], c5["pods"]
    rt, perm, st = w.check
    sub, lp = int(c5["subs"][0]), c5["lp"][0]
    o = c5.pop("o", None)  # (the eight stores below need the memory)
    if o is not None:
        o.close()
    del o
    engines = []

    def make(rank, nshards):
        e = aclgpu.Engine(w.schema, contexts=1)
        w.load(e)
        engines.append(e)
        return sharded.GpuShard(e, rank, nshards)

    def run(se):
        se._alloc(1 << 20)
        items = se.shard.e.make_items(rt, perm, w.res, st, "", w.subj)
        p, er = se.check_bulk_ids(items)
        bm = se.lookup_ids_batch(rt, perm, st, "", [sub])
        return p.cpu().numpy(), er.cpu().numpy(), bm.cpu().numpy()

    run.exchange = "alltoall"
    try:
        outs = sharded.run_logical_shards(8, make, run)
    finally:
        for e in engines:
            e.close()
    for p, er, bm in outs:
        assert np.array_equal(p, op) and np.array_equal(er, oe)
        bits = np.unpackbits(bm[0].view(np.uint8), bitorder="little")
        assert np.array_equal(bits[pods] == 1, lp == 2)
