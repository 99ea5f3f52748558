"""GPU parity of the SHARDED graph: G logical shards on one MI355X -- one Engine per shard,
the product's SPMD protocol (aclgpu/sharded.py) with an in-process communicator -- against the CPU oracle on
the unsharded graph and against the unsharded engine.  Real multi-GPU runs use the same code with RCCL."""
import numpy as np
import pytest

from oracle import orc
from tests.outcomes import outcome
from tests.test_oracle_cross import QUERIES, SCHEMA
from tests.test_sharded_gloo import CHAIN_SCHEMA, chain_case, random_tuples

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


def bits(row):
    return np.flatnonzero(np.unpackbits(row.cpu().numpy().view(np.uint8), bitorder="little")).astype(np.uint32)


@pytest.mark.parametrize("world,exchange", [(2, "allgather"), (8, "allgather"), (3, "alltoall"), (8, "alltoall")])
@pytest.mark.parametrize("name,kw", [("C2", dict(scale=0.05, batch=20000)), ("C3", dict(scale=0.05, batch=4000, power_users=8)),
                                     ("C4", dict(scale=0.02, batch=30000, n_user=20000))], ids=["C2", "C3", "C4"])
def test_sharded_workload_parity(name, kw, world, exchange, aclgpu):
    from aclgpu import sharded, workloads
    w = workloads.by_name(name, **kw)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rt, perm, st = w.check
    operms, oerrs = o.check_bulk_ids(rt, perm, w.res, st, "", w.subj)
    rng = np.random.default_rng(11)
    subs = [int(s) for s in rng.integers(0, w.nobjects[st], size=4)]
    if w.lookup_subjects is not None:
        subs += [int(s) for s in w.lookup_subjects[:4]]
    engines = []

    def make(rank, nshards):
        e = aclgpu.Engine(w.schema)
        w.load(e)
        engines.append(e)
        return sharded.GpuShard(e, rank, nshards)

    def run(se):
        items = se.shard.e.make_items(rt, perm, w.res, st, "", w.subj)
        p, er = se.check_bulk_ids(items)
        bm = se.lookup_ids_batch(rt, perm, st, "", subs)
        owners = {se.shard.owner_of_type(t) for t in w.nobjects if t != st}  # the types that hold relationships
        return p.cpu().numpy(), er.cpu().numpy(), bm.cpu(), se.exchanged_entries, owners, se.levels_last

    run.exchange = exchange
    try:
        outs = sharded.run_logical_shards(world, make, run)
    finally:
        for e in engines:
            e.close()
    for (p, er, bm, _x, _e, _l) in outs:  # every rank holds the full, identical answer
        assert np.array_equal(p, operms) and np.array_equal(er, oerrs)
        for i, s in enumerate(subs):
            assert np.array_equal(bits(bm[i]), np.sort(o.lookup_ids(rt, perm, st, "", s))), (name, world, s)
    if len(outs[0][4]) > 1:  # the type hash may put every relation-bearing type on one shard (few types, small world)
        assert sum(x[3] for x in outs) > 0, "nothing crossed a shard boundary"


def test_c5_mixed_stream_reduced(aclgpu):
    """BASELINE config 5 at reduced scale (200 k relationships, 8 logical shards): every Check batch and every Filter request
    of the mixed stream equals the oracle's answer."""
    from aclgpu import sharded, workloads
    w = workloads.c5(scale=0.002, batch=8000, n_lookups=6, n_user=4000)
    ops = workloads.c5_stream(w, 12)
    assert any(o == "C" for o in ops) and any(o != "C" for o in ops)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rt, perm, st = w.check
    operms, oerrs = o.check_bulk_ids(rt, perm, w.res, st, "", w.subj)
    engines = []

    def make(rank, nshards):
        e = aclgpu.Engine(w.schema)
        w.load(e)
        engines.append(e)
        return sharded.GpuShard(e, rank, nshards)

    def run(se):
        items = se.shard.e.make_items(rt, perm, w.res, st, "", w.subj)
        got = []
        for op in ops:
            if op == "C":
                p, er = se.check_bulk_ids(items)
                got.append((p.cpu().numpy(), er.cpu().numpy()))
            else:
                got.append(bits(se.lookup_ids_batch(rt, perm, st, "", [op[1]])[0]))
        return got

    try:
        outs = sharded.run_logical_shards(8, make, run)
    finally:
        for e in engines:
            e.close()
    for got in outs:
        for op, g in zip(ops, got):
            if op == "C":
                assert np.array_equal(g[0], operms) and np.array_equal(g[1], oerrs)
            else:
                assert np.array_equal(g, np.sort(o.lookup_ids(rt, perm, st, "", op[1]))), op


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_random_graphs_and_depth(world, aclgpu):
    """Cyclic nesting, arrows, permission-typed usersets (seeded random graphs) and the depth-50 chain whose every
    hop crosses shards; tiny export buffers so the redo path runs on the GPU too."""
    import random
    from aclgpu import sharded
    rng = random.Random(0xACE5 + world)
    cases = [(SCHEMA, random_tuples(rng, n), QUERIES,
              [("doc", "view", "user", "u0", ""), ("org", "view", "user", "u1", ""), ("group", "member", "group", "g0", "member"),
               ("doc", "view", "group", "g0", "member"), ("group", "manage", "user", "u2", "")]) for n in (0, 10, 24, 40)]
    for hops in (49, 50):
        cases.append((CHAIN_SCHEMA, chain_case(hops),
                      [("team", "t0", "member", "user", "deep", ""), ("team", "t0", "member", "user", "nobody", ""), ("crew", "c0", "member", "user", "deep", "")],
                      [("team", "member", "user", "deep", ""), ("crew", "member", "user", "deep", "")]))
    for ci, (schema, tuples, queries, lookups) in enumerate(cases):
        co = orc.Oracle(schema)
        for i in range(0, len(tuples), 500):
            co.write([(orc.OP_TOUCH, t) for t in tuples[i:i + 500]])
        want = [co.check(*q) for q in queries]
        engines = []

        def make(rank, nshards):
            e = aclgpu.Engine(schema)
            for i in range(0, len(tuples), 500):
                e.write([(aclgpu.OP_TOUCH, t) for t in tuples[i:i + 500]])
            for q in queries:  # identical interning order on every shard (the store is replicated)
                e.intern(q[0], q[1])
                e.intern(q[3], q[4])
            for l in lookups:
                e.intern(l[2], l[3])
            engines.append(e)
            return sharded.GpuShard(e, rank, nshards)

        def run(se):
            e = se.shard.e
            se._alloc(8)  # 8-entry export buffer: forces grow + redo
            items = np.zeros(len(queries), dtype=aclgpu.ITEM_DTYPE)
            for i, (rt, rid, pm, st, sid, sr) in enumerate(queries):
                items[i] = (e.type_id(rt), e.relation_id(rt, pm), e.find(rt, rid), e.type_id(st),
                            e.relation_id(st, sr) if sr else aclgpu.NO_RELATION, e.find(st, sid))
            p, er = se.check_bulk_ids(items)
            got = []
            for (rt, pm, st, sid, sr) in lookups:
                try:
                    bm = se.lookup_ids_batch(rt, pm, st, sr, [e.find(st, sid)])
                    got.append(("ok", {e.object_name(rt, int(b)) for b in bits(bm[0])}))
                except aclgpu.AclError as ex:  # a candidate's Check erred: EVERY shard fails the call, at the same point (they hold the same answers)
                    got.append(("err", ex.code))
            return list(zip(p.cpu().tolist(), er.cpu().tolist())), got

        run.exchange = "alltoall" if world % 2 else "allgather"
        try:
            outs = sharded.run_logical_shards(world, make, run)
        finally:
            for e in engines:
                e.close()
        for (got, lk) in outs:
            assert got == want, (ci, world, [(q, g, w_) for q, g, w_ in zip(queries, got, want) if g != w_][:5])
            for l, ids in zip(lookups, lk):
                assert ids == outcome(co.lookup, *l), (ci, world, l)


@pytest.mark.parametrize("world,xcap", [(2, 0), (8, 0), (5, 8)])
def test_native_loop_logical_shards(world, xcap, aclgpu, monkeypatch):
    """acl_shard_check_bulk -- the whole level loop inside libaclgpu.so, one fixed-capacity all-gather per level, decisions on the
    device -- on G logical shards of one GPU (the collective is a device-to-device copy between them; with RCCL it is
    ncclAllGather, same loop): answers equal the oracle's and the host-driven protocol's; xcap=8 forces grow-and-redo."""
    from aclgpu import sharded, workloads
    if xcap:
        monkeypatch.setenv("ACL_SHARD_XCAP", str(xcap))
    w = workloads.c4(scale=0.02, batch=30000, n_user=20000)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rt, perm, st = w.check
    operms, oerrs = o.check_bulk_ids(rt, perm, w.res, st, "", w.subj)
    engines = []

    def make(rank, nshards):
        e = aclgpu.Engine(w.schema, contexts=1)
        w.load(e)
        engines.append(e)
        return sharded.GpuShard(e, rank, nshards)

    def run(se):
        items = se.shard.e.make_items(rt, perm, w.res, st, "", w.subj)
        p, er, stats = se.check_bulk_ids_native(items)
        p2, er2, stats2 = se.check_bulk_ids_native(items)  # second batch: the burst is sized by the first one's depth
        return p.cpu().numpy(), er.cpu().numpy(), stats, p2.cpu().numpy(), er2.cpu().numpy(), stats2

    try:
        outs = sharded.run_logical_shards(world, make, run)
    finally:
        for e in engines:
            e.close()
    for p, er, stats, p2, er2, stats2 in outs:
        assert np.array_equal(p, operms) and np.array_equal(er, oerrs)
        assert np.array_equal(p2, operms) and np.array_equal(er2, oerrs)
        assert stats["levels"] == outs[0][2]["levels"] and stats["exchanges"] >= stats["levels"]
        assert stats2["host_syncs"] <= 3  # one per burst (the burst now covers the whole depth) + the final one
        if xcap:
            assert stats["retries"] >= 1 and stats["export_capacity"] > xcap


def test_native_loop_depth_chain(aclgpu):
    """The depth-50 chain whose every hop crosses shards, through the native loop: HAS at 50 dispatches, depth error at 51."""
    from aclgpu import sharded
    for hops in (49, 50):
        tuples = chain_case(hops)
        queries = [("team", "t0", "member", "user", "deep", ""), ("team", "t0", "member", "user", "nobody", ""), ("crew", "c0", "member", "user", "deep", "")]
        co = orc.Oracle(CHAIN_SCHEMA)
        co.write([(orc.OP_TOUCH, t) for t in tuples])
        want = [co.check(*q) for q in queries]
        engines = []

        def make(rank, nshards):
            e = aclgpu.Engine(CHAIN_SCHEMA, contexts=1)
            e.write([(aclgpu.OP_TOUCH, t) for t in tuples])
            for q in queries:
                e.intern(q[0], q[1])
                e.intern(q[3], q[4])
            engines.append(e)
            return sharded.GpuShard(e, rank, nshards)

        def run(se):
            e = se.shard.e
            items = np.zeros(len(queries), dtype=aclgpu.ITEM_DTYPE)
            for i, (rt, rid, pm, st, sid, sr) in enumerate(queries):
                items[i] = (e.type_id(rt), e.relation_id(rt, pm), e.find(rt, rid), e.type_id(st), aclgpu.NO_RELATION, e.find(st, sid))
            p, er, _s = se.check_bulk_ids_native(items)
            return list(zip(p.cpu().tolist(), er.cpu().tolist()))

        try:
            outs = sharded.run_logical_shards(3, make, run)
        finally:
            for e in engines:
                e.close()
        for got in outs:
            assert got == want, (hops, got, want)


def test_sharded_engine_refuses_unsharded_entry_points(aclgpu):
    from aclgpu import sharded
    with aclgpu.Engine(SCHEMA) as e:
        sharded.GpuShard(e, 0, 2)
        with pytest.raises(aclgpu.AclError) as ei:
            e.check("doc", "d0", "view", "user", "u0")
        assert ei.value.code == aclgpu.ERR_FAILED_PRECONDITION
        with pytest.raises(aclgpu.AclError):
            e.lookup("doc", "view", "user", "u0")


NCCL_WORLD1 = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [os.environ["ACL_ROOT"], os.path.join(os.environ["ACL_ROOT"], "spicedb-kubeapi-proxy_amd")]
import aclgpu
from aclgpu import sharded, workloads
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = workloads.c4(scale=0.02, batch=20000, n_user=20000)
e = aclgpu.Engine(w.schema); w.load(e)
items = e.make_items("pod", "view", w.res, "user", "", w.subj)
want = e.check_bulk_ids(items)
wl = e.lookup_ids_batch("pod", "view", "user", "", [int(w.subj[0]), int(w.subj[1])])[0]
se = sharded.ShardedEngine(sharded.GpuShard(e, 0, 1), sharded.TorchComm(device="cuda:0"))
p, er = se.check_bulk_ids(items)
bm = se.lookup_ids_batch("pod", "view", "user", "", [int(w.subj[0]), int(w.subj[1])])
ok = bool(np.array_equal(p.cpu().numpy(), want[0]) and np.array_equal(er.cpu().numpy(), want[1]) and np.array_equal(bm.cpu().numpy().view(np.uint32), wl))
# the collectives the protocol uses, with the dtypes it uses, on the engine's stream
with se.shard.stream():
    se.comm.all_gather(se.gather[:64], se.export[:64])
    h = torch.ones(8, dtype=torch.uint8, device="cuda"); se.comm.all_reduce_max(h)
    se.comm.broadcast(bm, 0)
# the native loop over the library's OWN RCCL communicator (ncclCommInitRank / ncclAllGather / ncclAllReduce inside libaclgpu.so)
pn, en, stn = se.check_bulk_ids_native(items)
ok_native = bool(np.array_equal(pn.cpu().numpy(), want[0]) and np.array_equal(en.cpu().numpy(), want[1]))
# ... and LookupResources through it (acl_shard_lookup_bulk_rccl)
bn, lstn = se.lookup_ids_batch_native("pod", "view", "user", "", [int(w.subj[0]), int(w.subj[1])])
ok_native = ok_native and bool(np.array_equal(bn.cpu().numpy().view(np.uint32), wl))
print(json.dumps({"ok": ok, "ok_native": ok_native, "native": stn, "native_lookup": lstn, "levels": se.levels_last}))
e.close(); dist.destroy_process_group()
"""


def test_protocol_over_rccl_world1(aclgpu, tmp_path):
    """The SPMD protocol with the REAL communicator (torch.distributed nccl == RCCL) on the one GPU a test box has:
    world_size 1 still runs every collective call, dtype and stream hand-off the multi-GPU path uses."""
    import json
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ACL_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", NCCL_WORLD1], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["ok_native"], out
    assert out["native"]["exchanges"] >= out["native"]["levels"] >= 1


NCCL_WORLD_N = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [os.environ["ACL_ROOT"], os.path.join(os.environ["ACL_ROOT"], "spicedb-kubeapi-proxy_amd")]
import aclgpu
from aclgpu import sharded, workloads
from oracle import orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
w = workloads.c4(scale=0.02, batch=20000, n_user=20000)
o = orc.Oracle(w.schema); w.load(o); o.freeze()
want = o.check_bulk_ids("pod", "view", w.res, "user", "", w.subj)
subs = [int(w.subj[0]), int(w.subj[1]), 7]
e = aclgpu.Engine(w.schema, device=rank); w.load(e)
se = sharded.ShardedEngine(sharded.GpuShard(e, rank, world), sharded.TorchComm(device=f"cuda:{rank}"))
items = e.make_items("pod", "view", w.res, "user", "", w.subj)
res = {}
for name in ("first", "planned"):  # the second batch runs on the first one's plan
    p, er, st = se.check_bulk_ids_native(items)   # ncclSend / ncclRecv (all-to-all) + ncclAllGather of headers + ncclAllReduce, between two DEVICES
    res[name] = {"ok": bool(np.array_equal(p.cpu().numpy(), want[0]) and np.array_equal(er.cpu().numpy(), want[1])), "stats": st}
bm, lst = se.lookup_ids_batch_native("pod", "view", "user", "", subs)
rows = bm.cpu().numpy().view(np.uint32)
ok_l = all(np.array_equal(np.flatnonzero(np.unpackbits(rows[i].view(np.uint8), bitorder="little")), np.sort(o.lookup_ids("pod", "view", "user", "", s))) for i, s in enumerate(subs))
ph, eh = se.check_bulk_ids(items)  # the host-driven protocol over torch.distributed, same communicator family
ok_h = bool(np.array_equal(ph.cpu().numpy(), want[0]) and np.array_equal(eh.cpu().numpy(), want[1]))
print(json.dumps({"rank": rank, "check": res, "lookup_ok": bool(ok_l), "lookup": lst, "host_protocol_ok": ok_h, "local_relationships": e.stats()["snapshot_edges_local"]}))
e.close(); dist.destroy_process_group()
"""


def test_native_loop_between_two_devices(aclgpu, tmp_path):
    """Frontier entries moving between two DEVICES over the library's own RCCL communicator ("no test has moved a frontier
    entry between two devices").  Needs >= 2 visible GPUs: the development and round-end test boxes have one, so this SKIPS LOUDLY there;
    on a multi-GPU node it runs one process per GPU (the deployment shape) and checks every answer of both ranks against the oracle."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import warnings
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        warnings.warn(f"SHARDED EXCHANGE BETWEEN DEVICES NOT EXERCISED: {ndev} GPU visible, the two-device test needs 2")
        pytest.skip(f"{ndev} GPU visible: frontier entries between two devices cannot be exercised here")
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   ACL_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        procs.append(subprocess.Popen([sys.executable, "-c", NCCL_WORLD_N], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se_ = p.communicate(timeout=600)
        assert p.returncode == 0, so + se_
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
    for o_ in outs:
        assert o_["check"]["first"]["ok"] and o_["check"]["planned"]["ok"] and o_["lookup_ok"] and o_["host_protocol_ok"], o_
        assert o_["check"]["first"]["stats"]["entries_exchanged"] > 0
    assert sum(o_["local_relationships"] for o_ in outs) > 0


@pytest.mark.parametrize("world,a2a", [(2, True), (8, True), (5, False)])
def test_native_loop_lookup_all_to_all_and_plan(world, a2a, aclgpu, monkeypatch):
    """The native loop's three additions of round 3, on G logical shards of one GPU: (1) LookupResources inside the library
    (acl_shard_lookup_bulk: VISIT -> exchange -> import -> EXPAND pairs, rows MAX-reduced from the resource type's owner) equals the oracle's sets
    and the host-driven protocol's rows; (2) Check with per-destination blocks through the communicator's all_to_all (or, without one, the
    all-gather form) gives the oracle's answers; (3) the PLAN: a second identical batch exchanges entry blocks only on the levels the first one
    exported on, and a batch of another shape that exports where none was planned is redone (retries >= 1) -- same answers either way."""
    from aclgpu import sharded, workloads
    monkeypatch.setenv("ACL_SHARD_A2A", "1" if a2a else "0")
    w = workloads.c4(scale=0.02, batch=20000, n_user=20000)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rt, perm, st = w.check
    operms, oerrs = o.check_bulk_ids(rt, perm, w.res, st, "", w.subj)
    rng = np.random.default_rng(3)
    users = rng.integers(0, w.nobjects["user"], size=12).astype(np.uint32)
    groups = rng.integers(0, w.nobjects["group"], size=6).astype(np.uint32)
    # a batch of ANOTHER shape: namespace checks whose subjects are groups (userset subjects: the first level already crosses to the group shard)
    ns_res = rng.integers(0, w.nobjects["namespace"], size=3000).astype(np.uint32)
    ns_sub = rng.integers(0, w.nobjects["user"], size=3000).astype(np.uint32)
    nperms, nerrs = o.check_bulk_ids("namespace", "view", ns_res, "user", "", ns_sub)
    engines = []

    def make(rank, nshards):
        e = aclgpu.Engine(w.schema, contexts=1)
        w.load(e)
        engines.append(e)
        return sharded.GpuShard(e, rank, nshards)

    def run(se):
        e = se.shard.e
        items = e.make_items(rt, perm, w.res, st, "", w.subj)
        p1, e1, s1 = se.check_bulk_ids_native(items)
        p2, e2, s2 = se.check_bulk_ids_native(items)
        nitems = e.make_items("namespace", "view", ns_res, "user", "", ns_sub)
        p3, e3, s3 = se.check_bulk_ids_native(nitems)
        lk = {}
        for key, (lrt, lperm, lst, lsrel, subs) in {"pod/user": ("pod", "view", "user", "", users), "group/user": ("group", "member", "user", "", users),
                                                      "pod/group": ("pod", "view", "group", "member", groups)}.items():
            b1, ls1 = se.lookup_ids_batch_native(lrt, lperm, lst, lsrel, subs)
            b2, ls2 = se.lookup_ids_batch_native(lrt, lperm, lst, lsrel, subs)
            host = se.lookup_ids_batch(lrt, lperm, lst, lsrel, subs)  # the host-driven step protocol
            lk[key] = (b1.cpu().numpy(), b2.cpu().numpy(), host.cpu().numpy(), ls1, ls2)
        return (p1.cpu().numpy(), e1.cpu().numpy(), s1, p2.cpu().numpy(), e2.cpu().numpy(), s2, p3.cpu().numpy(), e3.cpu().numpy(), s3, lk)

    try:
        outs = sharded.run_logical_shards(world, make, run)
    finally:
        for e in engines:
            e.close()
    for p1, e1, s1, p2, e2, s2, p3, e3, s3, lk in outs:
        assert np.array_equal(p1, operms) and np.array_equal(e1, oerrs)
        assert np.array_equal(p2, operms) and np.array_equal(e2, oerrs)
        assert np.array_equal(p3, nperms) and np.array_equal(e3, nerrs)
        assert s1["data_exchanges"] == s1["exchanges"]  # no plan yet: entries on every level
        assert s2["retries"] == 0 and s2["data_exchanges"] <= s2["exchanges"]
        if world > 2:  # (with two shards of four types the group levels may still cross; with more the nested levels stay home)
            assert s2["data_exchanges"] < s2["exchanges"], s2
        for key, (b1, b2, host, ls1, ls2) in lk.items():
            assert np.array_equal(b1, b2) and np.array_equal(b1, host), key
            assert ls2["retries"] == 0
    lrt = {"pod/user": ("pod", "view", "user", "", users), "group/user": ("group", "member", "user", "", users), "pod/group": ("pod", "view", "group", "member", groups)}
    for key, (lrt_, lperm, lst, lsrel, subs) in lrt.items():
        rows = outs[0][9][key][0]
        for i, s_ in enumerate(subs):
            want = np.sort(o.lookup_ids(lrt_, lperm, lst, lsrel, int(s_)))
            got = np.flatnonzero(np.unpackbits(rows[i].view(np.uint8), bitorder="little")).astype(np.uint32)
            assert np.array_equal(got, want), (key, int(s_), got.size, want.size)


IPC_WORLD_N = r"""
import os, sys, json, types
import numpy as np, torch
sys.path[:0] = [os.environ["ACL_ROOT"], os.path.join(os.environ["ACL_ROOT"], "spicedb-kubeapi-proxy_amd")]
import aclgpu
from aclgpu import sharded, workloads
from oracle import orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
w = workloads.c4(scale=0.02, batch=20000, n_user=20000)
o = orc.Oracle(w.schema); w.load(o); o.freeze()
want = o.check_bulk_ids("pod", "view", w.res, "user", "", w.subj)
rng = np.random.default_rng(3)
users = rng.integers(0, w.nobjects["user"], size=12).astype(np.uint32)
groups = rng.integers(0, w.nobjects["group"], size=6).astype(np.uint32)
ns_res = rng.integers(0, w.nobjects["namespace"], size=3000).astype(np.uint32)
ns_sub = rng.integers(0, w.nobjects["user"], size=3000).astype(np.uint32)
nwant = o.check_bulk_ids("namespace", "view", ns_res, "user", "", ns_sub)
e = aclgpu.Engine(w.schema, contexts=1); w.load(e)
# the communicator: one window of device memory per PROCESS, mapped by the peer through hipIpcOpenMemHandle; a barrier in POSIX shared memory
ipc = sharded.IpcNative(os.environ["IPC_NAME"], rank, world, device=0, window_bytes=int(os.environ["IPC_WINDOW"]), deadline_s=120,
                        with_all_to_all=os.environ["ACL_SHARD_A2A"] == "1")
se = sharded.ShardedEngine(sharded.GpuShard(e, rank, world), types.SimpleNamespace(rank=rank, world=world), native=ipc)
ipc.barrier()  # both processes have their graph in HBM
items = e.make_items("pod", "view", w.res, "user", "", w.subj)
res = {}
for name in ("first", "planned"):  # the second batch runs on the first one's plan
    p, er, st = se.check_bulk_ids_native(items)
    res[name] = {"ok": bool(np.array_equal(p.cpu().numpy(), want[0]) and np.array_equal(er.cpu().numpy(), want[1])), "stats": st}
p, er, st = se.check_bulk_ids_native(e.make_items("namespace", "view", ns_res, "user", "", ns_sub))  # another shape: exports where none were planned
res["other_shape"] = {"ok": bool(np.array_equal(p.cpu().numpy(), nwant[0]) and np.array_equal(er.cpu().numpy(), nwant[1])), "stats": st}
lk = {}
for key, (lrt, lperm, lst, lsrel, subs) in {"pod/user": ("pod", "view", "user", "", users), "group/user": ("group", "member", "user", "", users),
                                              "pod/group": ("pod", "view", "group", "member", groups)}.items():
    out = []
    for _round in range(2):
        bm, lstat = se.lookup_ids_batch_native(lrt, lperm, lst, lsrel, subs)
        rows = bm.cpu().numpy().view(np.uint32)
        ok = all(np.array_equal(np.flatnonzero(np.unpackbits(rows[i].view(np.uint8), bitorder="little")), np.sort(o.lookup_ids(lrt, lperm, lst, lsrel, int(s))))
                 for i, s in enumerate(subs))
        out.append({"ok": bool(ok), "stats": lstat, "ids": int(sum(int(np.unpackbits(rows[i].view(np.uint8)).sum()) for i in range(len(subs))))})
    lk[key] = out
print(json.dumps({"rank": rank, "pid": os.getpid(), "check": res, "lookup": lk, "ipc": ipc.stats(), "local_relationships": e.stats()["snapshot_edges_local"]}))
ipc.barrier()  # nobody unmaps a window a peer may still read
ipc.close(); e.close()
"""


@pytest.mark.parametrize("a2a,window", [(True, 8 << 20), (False, 64 << 10)])
def test_native_loops_between_two_processes_on_one_gpu(a2a, window, aclgpu, tmp_path):
    """engine_shard_native.cpp across a PROCESS boundary on the one GPU a test box has.  Two processes share device 0, each
    holds one shard; the communicator (tools/ipc_comm.hip) is an acl_shard_comm_t over hipIpcGetMemHandle / hipIpcOpenMemHandle windows and a
    shared-memory barrier -- RCCL refuses two ranks on one device, the callbacks interface does not.  What the in-process thread double cannot
    cover and this does: separate address spaces and HIP contexts, streams ordered against a foreign process, bursts and host_syncs with the
    peer running at its own pace.  Check (first batch, planned batch, a batch of another shape that is redone) and LookupResources (three subject
    classes, twice) equal the oracle's on BOTH ranks; frontier entries really crossed (entries_exchanged > 0, bytes pulled out of the foreign
    window > 0).  a2a=True: per-destination blocks for Check and -- new in round 4 -- for LookupResources (states go to the shards that hold
    parent rows for them); a2a=False with a 64 KiB window: the all-gather form, every 1 MiB export block in sixteen slices."""
    import json
    import os
    import subprocess
    import sys
    name = f"/aclipc-test-{os.getpid()}-{int(a2a)}"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0", IPC_NAME=name, IPC_WINDOW=str(window), ACL_SHARD_A2A="1" if a2a else "0",
                   ACL_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        procs.append(subprocess.Popen([sys.executable, "-c", IPC_WORLD_N], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se_ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, so[-3000:] + se_[-3000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
    assert outs[0]["pid"] != outs[1]["pid"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)  # (scratch; what a run on the GPU box leaves there is copied to profiles/ by hand)
    with open(os.path.join(root, "gpurun_out", f"ipc_two_processes_{'a2a' if a2a else 'allgather_sliced'}.json"), "w") as f:
        json.dump(outs, f, indent=1)
    for o_ in outs:
        for k in ("first", "planned", "other_shape"):
            assert o_["check"][k]["ok"], (o_["rank"], k, o_["check"][k]["stats"])
        assert o_["check"]["first"]["stats"]["entries_exchanged"] > 0 and o_["check"]["first"]["stats"]["data_exchanges"] == o_["check"]["first"]["stats"]["exchanges"]
        assert o_["check"]["planned"]["stats"]["retries"] == 0
        for key, rounds in o_["lookup"].items():
            assert all(r_["ok"] for r_ in rounds), (o_["rank"], key, [r_["stats"] for r_ in rounds])
            assert rounds[0]["ids"] == rounds[1]["ids"] and rounds[1]["stats"]["retries"] == 0
        assert o_["lookup"]["pod/user"][0]["stats"]["entries_exchanged"] > 0 and o_["lookup"]["pod/user"][0]["ids"] > 0
        assert o_["ipc"]["foreign_bytes"] > 0 and o_["ipc"]["collectives"] > 10
    # the two ranks saw the same control records: same levels, same exchanges
    for k in ("first", "planned", "other_shape"):
        assert outs[0]["check"][k]["stats"]["levels"] == outs[1]["check"][k]["stats"]["levels"]
        assert outs[0]["check"][k]["stats"]["entries_exchanged"] == outs[1]["check"][k]["stats"]["entries_exchanged"]
    assert all(o_["local_relationships"] > 0 for o_ in outs)  # both shards hold rows


@pytest.mark.parametrize("world", [2, 5, 8])
def test_native_loop_combine_schema_bans_workload(world, aclgpu):
    """schemas with `-`, `&`, `T:*` and a non-monotone userset subject (`group#active`) on the SHARDED graph.  The native
    Check loop hands out leaf cells from per-shard ranges of one global cell space, maxes the whole cell space behind the walk, gathers the
    shards' combine nodes and resolves all of them on every shard: `view` = (...) - banned, `strict` = creator & namespace->view, `loose`
    (pure union, same kernels) and `group#active` equal the oracle's on 2 / 5 / 8 logical shards, twice (the second batch runs on the plan
    the first one left)."""
    from aclgpu import sharded
    from tests.test_combine_gpu import SCHEMA_BANS, bans_graph, load_numeric
    E, n = bans_graph(11, n_user=2000, n_group=300, n_ns=100, n_pod=8000)
    co = orc.Oracle(SCHEMA_BANS)
    load_numeric(co, E)
    co.freeze()
    rng = np.random.default_rng(3)
    B = 30000
    res = rng.integers(0, n["pod"], size=B).astype(np.uint32)
    sub = rng.integers(0, n["user"], size=B).astype(np.uint32)
    creators = dict(zip(E[8][4].tolist(), E[8][5].tolist()))
    for i in range(0, B, 3):
        sub[i] = creators[int(res[i])]
    gres = rng.integers(0, n["group"], size=5000).astype(np.uint32)
    gsub = rng.integers(0, n["user"], size=5000).astype(np.uint32)
    want = {p: co.check_bulk_ids_mt(8, "pod", p, res, "user", "", sub) for p in ("view", "strict", "loose")}
    want_g = co.check_bulk_ids_mt(8, "group", "active", gres, "user", "", gsub)
    assert all(0 < int((w[0] == 2).sum()) < B for w in want.values())
    lsubs = np.array([int(x) for x in rng.integers(0, n["user"], size=5)] + [int(sub[0])], dtype=np.uint32)
    want_lk = {p: [np.sort(co.lookup_ids("pod", p, "user", "", int(s_))) for s_ in lsubs] for p in ("view", "strict")}
    want_lk["active"] = [np.sort(co.lookup_ids("group", "active", "user", "", int(s_))) for s_ in lsubs]
    assert sum(x.size for x in want_lk["view"]) > 0
    engines = []

    def make(rank, nshards):
        e = aclgpu.Engine(SCHEMA_BANS, contexts=1)
        load_numeric(e, E)
        engines.append(e)
        return sharded.GpuShard(e, rank, nshards)

    def run(se):
        e = se.shard.e
        out = {}
        for _round in range(2):
            for p in ("view", "strict", "loose"):
                pp, er, stats = se.check_bulk_ids_native(e.make_items("pod", p, res, "user", "", sub))
                out[(p, _round)] = (pp.cpu().numpy(), er.cpu().numpy(), stats)
        pp, er, stats = se.check_bulk_ids_native(e.make_items("group", "active", gres, "user", "", gsub))
        out["g"] = (pp.cpu().numpy(), er.cpu().numpy(), stats)
        # Filter over non-monotone permissions: the sharded reverse walk's candidates + ONE sharded Check
        for p in ("view", "strict"):
            bm, lstat = se.lookup_ids_batch_native("pod", p, "user", "", lsubs)
            out[("lk", p)] = ([bits(bm[i]) for i in range(len(lsubs))], lstat)
        bm, lstat = se.lookup_ids_batch_native("group", "active", "user", "", lsubs)
        out[("lk", "active")] = ([bits(bm[i]) for i in range(len(lsubs))], lstat)
        return out

    try:
        outs = sharded.run_logical_shards(world, make, run)
    finally:
        for e in engines:
            e.close()
    moved = 0
    for out in outs:
        for p in ("view", "strict", "loose"):
            for r in range(2):
                got = out[(p, r)]
                assert np.array_equal(got[0], want[p][0]) and np.array_equal(got[1], want[p][1]), (world, p, r, int((got[0] != want[p][0]).sum()))
                moved += got[2]["entries_exchanged"]
        assert np.array_equal(out["g"][0], want_g[0]) and np.array_equal(out["g"][1], want_g[1])
        for p in ("view", "strict", "active"):
            for got_ids, want_ids in zip(out[("lk", p)][0], want_lk[p]):
                assert np.array_equal(got_ids, want_ids), (world, p, got_ids.size, want_ids.size)
    assert moved > 0  # (sub-walks of leaf cells did cross shards)


def test_native_loop_combine_schema_named_cases(aclgpu):
    """The combine reference case (tests/ref_cases.py: precedence, wildcards in positive and subtracted operands, arrows into non-monotone
    permissions, intersection arrows over several folders, depth errors on both sides of `&` / `-` through a 60-long chain) on 3 logical
    shards: every answer equals the C oracle's AND the independent Python oracle's."""
    from aclgpu import sharded
    from oracle.pyoracle import PyOracle
    from tests import ref_cases
    from tests.test_combine_gpu import PY2C
    case = ref_cases._combine_case()
    co, po = orc.Oracle(case["schema"]), PyOracle(case["schema"])
    co.write([(orc.OP_TOUCH, r) for r in case["relationships"]])
    for r in case["relationships"]:
        po.touch(*r)
    queries = case["checks"]
    want = [co.check(*q) for q in queries]
    assert want == [PY2C[po.check(*q)] for q in queries]
    assert any(w[1] for w in want) and any(w[0] == 2 for w in want)
    engines = []

    def make(rank, nshards):
        e = aclgpu.Engine(case["schema"], contexts=1)
        rels = case["relationships"]
        for i in range(0, len(rels), 1000):
            e.write([(aclgpu.OP_TOUCH, r) for r in rels[i:i + 1000]])
        for q in queries:  # (unknown objects get ids too: every shard interns the same names in the same order)
            e.intern(q[0], q[1])
            e.intern(q[3], q[4])
        engines.append(e)
        return sharded.GpuShard(e, rank, nshards)

    def run(se):
        e = se.shard.e
        items = np.zeros(len(queries), dtype=aclgpu.ITEM_DTYPE)
        for i, (rt, rid, pm, st, sid, sr) in enumerate(queries):
            items[i] = (e.type_id(rt), e.relation_id(rt, pm), e.find(rt, rid), e.type_id(st), e.relation_id(st, sr) if sr else aclgpu.NO_RELATION, e.find(st, sid))
        p, er, _s = se.check_bulk_ids_native(items)
        return list(zip(p.cpu().tolist(), er.cpu().tolist()))

    try:
        outs = sharded.run_logical_shards(3, make, run)
    finally:
        for e in engines:
            e.close()
    for got in outs:
        bad = [(q, g, w) for q, g, w in zip(queries, got, want) if g != w]
        assert not bad, bad[:5]


IPC_WORLD_COMBINE = r"""
import os, sys, json, types
import numpy as np, torch
sys.path[:0] = [os.environ["ACL_ROOT"], os.path.join(os.environ["ACL_ROOT"], "spicedb-kubeapi-proxy_amd")]
import aclgpu
from aclgpu import sharded
from oracle import orc
from tests.test_combine_gpu import SCHEMA_BANS, bans_graph, load_numeric
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
E, n = bans_graph(11, n_user=2000, n_group=300, n_ns=100, n_pod=8000)
o = orc.Oracle(SCHEMA_BANS); load_numeric(o, E); o.freeze()
rng = np.random.default_rng(3)
B = 20000
res = rng.integers(0, n["pod"], size=B).astype(np.uint32)
sub = rng.integers(0, n["user"], size=B).astype(np.uint32)
creators = dict(zip(E[8][4].tolist(), E[8][5].tolist()))
for i in range(0, B, 3):
    sub[i] = creators[int(res[i])]
subs = np.array([int(x) for x in rng.integers(0, n["user"], size=4)] + [int(sub[0])], dtype=np.uint32)
e = aclgpu.Engine(SCHEMA_BANS, contexts=1); load_numeric(e, E)
ipc = sharded.IpcNative(os.environ["IPC_NAME"], rank, world, device=0, window_bytes=int(os.environ["IPC_WINDOW"]), deadline_s=120, with_all_to_all=os.environ["ACL_SHARD_A2A"] == "1")
se = sharded.ShardedEngine(sharded.GpuShard(e, rank, world), types.SimpleNamespace(rank=rank, world=world), native=ipc)
ipc.barrier()
out = {"check": {}, "lookup": {}}
for perm in ("view", "strict", "loose"):
    want = o.check_bulk_ids_mt(4, "pod", perm, res, "user", "", sub)
    for rnd in range(2):
        p, er, st = se.check_bulk_ids_native(e.make_items("pod", perm, res, "user", "", sub))
        out["check"][f"{perm}/{rnd}"] = {"ok": bool(np.array_equal(p.cpu().numpy(), want[0]) and np.array_equal(er.cpu().numpy(), want[1])), "stats": st, "has": int((want[0] == 2).sum())}
for perm in ("view", "strict"):
    bm, lstat = se.lookup_ids_batch_native("pod", perm, "user", "", subs)
    rows = bm.cpu().numpy().view(np.uint32)
    ok = all(np.array_equal(np.flatnonzero(np.unpackbits(rows[i].view(np.uint8), bitorder="little")), np.sort(o.lookup_ids("pod", perm, "user", "", int(s)))) for i, s in enumerate(subs))
    out["lookup"][perm] = {"ok": bool(ok), "stats": lstat, "ids": int(sum(int(np.unpackbits(rows[i].view(np.uint8)).sum()) for i in range(len(subs))))}
print(json.dumps({"rank": rank, "pid": os.getpid(), **out, "ipc": ipc.stats(), "local_relationships": e.stats()["snapshot_edges_local"]}))
ipc.barrier()
ipc.close(); e.close()
"""


@pytest.mark.parametrize("a2a", [True, False])
def test_native_loops_combine_schema_between_two_processes(a2a, aclgpu):
    """The "banned users" schema (`-`, `&`, `T:*`, a non-monotone userset subject) on TWO PROCESSES sharing the GPU, one shard each, over the hipIpc
    communicator: Check (three permissions, twice) and LookupResources (candidates + one sharded Check) equal the oracle's on both ranks; leaf
    cells' sub-walks crossed the process boundary (entries_exchanged > 0)."""
    import json
    import os
    import subprocess
    import sys
    name = f"/aclipc-cmb-{os.getpid()}-{int(a2a)}"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0", IPC_NAME=name, IPC_WINDOW=str(8 << 20), ACL_SHARD_A2A="1" if a2a else "0",
                   ACL_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        procs.append(subprocess.Popen([sys.executable, "-c", IPC_WORLD_COMBINE], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se_ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, so[-3000:] + se_[-3000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    assert outs[0]["pid"] != outs[1]["pid"]
    for o_ in outs:
        assert all(v["ok"] for v in o_["check"].values()), (o_["rank"], {k: v["stats"] for k, v in o_["check"].items() if not v["ok"]})
        assert all(v["ok"] for v in o_["lookup"].values()), (o_["rank"], o_["lookup"])
        assert o_["check"]["view/0"]["stats"]["entries_exchanged"] > 0 and 0 < o_["check"]["view/0"]["has"] < 20000
        assert o_["lookup"]["view"]["ids"] > 0 and o_["ipc"]["foreign_bytes"] > 0
    assert all(o_["local_relationships"] > 0 for o_ in outs)
