"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP frontier kernels, called
through the C ABI, against the CPU oracle on the same inputs.  Bit-exact bar:
identical permissionship AND identical per-item error code for every request,
identical id SETS for every LookupResources."""
import numpy as np
import pytest

from oracle import orc
from tests import kat_runner
from tests.test_oracle_cross import QUERIES, SCHEMA, tuples_strategy

pytestmark = pytest.mark.gpu

KATS = kat_runner.load_kats()


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_engine_kat(kat, aclgpu):
    """Golden vectors of the reference's own tests through the GPU engine."""
    schema, rels = kat_runner.kat_schema(kat)
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        kat_runner.run_kat(kat, e)


def test_client_mirror_reads_like_reference_tests(aclgpu):
    """KAT-1 written the way pkg/authz/distributedtx/workflow_test.go:79-119 does it, through the
    PermissionsServiceClient mirror; plus the post-filter rule of postfilter.go:144-178."""
    from aclgpu import client as v1
    b = kat_runner.load_bootstrap()
    with aclgpu.Engine(b["schema"], "\n".join(b["relationships"])) as e:
        c = v1.PermissionsServiceClient(e)
        c.WriteRelationships([v1.RelationshipUpdate(v1.OPERATION_CREATE, v1.Relationship(
            v1.ObjectReference("namespace", "my_object_meta"), "creator", v1.SubjectReference(v1.ObjectReference("user", "janedoe"))))])
        r = c.CheckPermission(v1.CheckPermissionRequest(v1.ObjectReference("namespace", "my_object_meta"), "view",
                                                        v1.SubjectReference(v1.ObjectReference("user", "janedoe"))))
        assert r.permissionship == v1.PERMISSIONSHIP_HAS_PERMISSION
        with pytest.raises(aclgpu.AclError) as ei:  # pkg/proxy/options_test.go:101-102
            c.CheckPermission(v1.CheckPermissionRequest())
        assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT
        items = [v1.CheckPermissionRequest(v1.ObjectReference("namespace", n), "view", v1.SubjectReference(v1.ObjectReference("user", u)))
                 for n, u in [("my_object_meta", "janedoe"), ("my_object_meta", "rakis"), ("spicedb-kubeapi-proxy", "rakis"), ("nope", "janedoe")]]
        resp = c.CheckBulkPermissions(items)
        assert [v1.is_allowed(p) for p in resp.pairs] == [True, False, True, False]
        got = {r.resource_object_id for r in c.LookupResources(v1.LookupResourcesRequest("namespace", "view", v1.SubjectReference(v1.ObjectReference("user", "rakis"))))}
        assert got == {"spicedb-kubeapi-proxy"}
        assert c.DeleteRelationships(v1.RelationshipFilter("namespace", "my_object_meta")).relationships_deleted_count == 1
        assert list(c.ReadRelationships(v1.RelationshipFilter("namespace", "my_object_meta"))) == []


def test_depth_limit_chain(aclgpu):
    """Dispatch depth 50 (pkg/spicedb/spicedb.go:34): HAS while <= 50 dispatches, error beyond -- same as the oracle."""
    schema = "definition user {}\ndefinition group { relation member: user | group#member }"
    n = 60
    rels = [f"group:g{i}#member@group:g{i+1}#member" for i in range(n)] + [f"group:g{n}#member@user:deep"]
    o = orc.Oracle(schema)
    for i in range(0, len(rels), 500):
        o.write([(orc.OP_TOUCH, r) for r in rels[i:i + 500]])
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        for k in range(0, n + 1):
            assert e.check("group", f"g{k}", "member", "user", "deep") == o.check("group", f"g{k}", "member", "user", "deep"), k
        assert e.check("group", f"g{n - 49}", "member", "user", "deep") == (2, 0)
        assert e.check("group", f"g{n - 50}", "member", "user", "deep") == (0, aclgpu.ERR_DEPTH)
        assert e.lookup("group", "member", "user", "deep") == o.lookup("group", "member", "user", "deep")


def test_branching_cycles_stay_polynomial(aclgpu):
    """Group nesting with BRANCHING cycles (g0 -> {g0, g1}, g1 -> g0: found by the hypothesis test below) doubles the number of
    pending sub-checks per dispatch level for 50 levels unless identical (request, state, level) entries are merged.  The engine
    merges them once a pass outgrows its frontier; the answers are the oracle's (which memoises on the same key): depth error for
    subjects that are nowhere, HAS where a member is reachable."""
    schema = "definition user {}\ndefinition group { relation member: user | group#member }"
    rels = ["group:g0#member@group:g0#member", "group:g0#member@group:g1#member", "group:g1#member@group:g0#member",
            "group:g1#member@group:g2#member", "group:g2#member@group:g0#member", "group:g2#member@group:g1#member", "group:g2#member@user:deep"]
    o = orc.Oracle(schema)
    o.write([(orc.OP_TOUCH, r) for r in rels])
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        for g in ("g0", "g1", "g2"):
            for u in ("deep", "nobody"):
                assert e.check("group", g, "member", "user", u) == o.check("group", g, "member", "user", u), (g, u)
        assert e.check("group", "g0", "member", "user", "nobody")[0] != 2
        qs = [("group", g, "member", "user", u, "") for g in ("g0", "g1", "g2") for u in ("deep", "nobody")] * 50
        perms, errs = e.check_bulk(qs)
        assert list(zip(perms, errs)) == [o.check(*q) for q in qs]
        assert e.stats()["overflow_retries"] >= 1  # (the merging pass was what answered)
        # more requests than one merging pass holds (its key has 14 request bits): the batch is answered in slices
        import numpy as np
        base = e.make_items("group", "member", np.zeros(1, dtype=np.uint32), "user", "", np.zeros(1, dtype=np.uint32))
        ids = {g: int(e.intern("group", g)) for g in ("g0", "g1", "g2")}
        us = {u: int(e.intern("user", u)) for u in ("deep", "nobody")}
        combos = [(g, u) for g in ("g0", "g1", "g2") for u in ("deep", "nobody")]
        big = np.repeat(base, 20000)
        for i, (g, u) in enumerate(combos):
            big["resource_id"][i::6] = ids[g]
            big["subject_id"][i::6] = us[u]
        bp, be = e.check_bulk_ids(big)
        for i, (g, u) in enumerate(combos):
            want = o.check("group", g, "member", "user", u)
            assert set(zip(bp[i::6].tolist(), be[i::6].tolist())) == {want}, (g, u)


def test_engine_matches_oracle_hypothesis(aclgpu):
    """Random small graphs incl. cyclic group nesting, arrows, usersets with permissions as subject
    relations: every query's (permissionship, error) and every lookup set equal the oracle's."""
    from hypothesis import given, settings

    eng = aclgpu.Engine(SCHEMA)
    subjects = [("user", "u0", ""), ("group", "g0", "member"), ("group", "g1", "manage")]

    @settings(max_examples=40, deadline=None)
    @given(tuples_strategy())
    def run(tuples):
        eng.load_bootstrap(SCHEMA)
        co = orc.Oracle(SCHEMA)
        tuples = list(dict.fromkeys(tuples))
        if tuples:
            co.write([(orc.OP_TOUCH, t) for t in tuples])
            eng.write([(aclgpu.OP_TOUCH, t) for t in tuples])
        perms, errs = eng.check_bulk(QUERIES)
        want = [co.check(*q) for q in QUERIES]
        assert list(zip(perms, errs)) == want
        for s in subjects:
            for rt, p in [("doc", "view"), ("org", "view"), ("group", "member"), ("group", "manage"), ("doc", "nothing")]:
                assert eng.lookup(rt, p, *s) == co.lookup(rt, p, *s), (rt, p, s)

    run()
    eng.close()


WORKLOADS = [("C1", {}), ("C2", dict(scale=0.05, batch=20000)), ("C3", dict(scale=0.05, batch=5000, power_users=8)),
             ("C4", dict(scale=0.02, batch=30000, n_user=20000))]


@pytest.mark.parametrize("name,kw", WORKLOADS, ids=[w[0] for w in WORKLOADS])
def test_workload_parity(name, kw, aclgpu):
    """BASELINE configs at reduced scale: interned-id bulk Check and Filter bitmaps vs the oracle."""
    from aclgpu import workloads
    w = workloads.by_name(name, **kw)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        rt, perm, st = w.check
        items = e.make_items(rt, perm, w.res, st, "", w.subj)
        perms, errs = e.check_bulk_ids(items)
        operms, oerrs = o.check_bulk_ids(rt, perm, w.res, st, "", w.subj)
        assert np.array_equal(perms, operms)
        assert np.array_equal(errs, oerrs)
        assert 0 < (perms == 2).sum() < perms.size or name == "C4"
        # device-resident entry point gives the same bytes
        import torch
        d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
        d_perm = torch.zeros(items.size, dtype=torch.uint8, device="cuda")
        d_err = torch.zeros(items.size, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        e.check_bulk_ids_device(d_items.data_ptr(), items.size, d_perm.data_ptr(), d_err.data_ptr())
        e.sync()
        assert np.array_equal(d_perm.cpu().numpy(), operms) and np.array_equal(d_err.cpu().numpy(), oerrs)
        # Filter: LookupResources bitmaps == {id : oracle check == HAS}
        rng = np.random.default_rng(7)
        subs = list(rng.integers(0, w.nobjects[st], size=6))
        if w.lookup_subjects is not None:
            subs += list(w.lookup_subjects[:6])
        bms, counts = e.lookup_ids_batch(rt, perm, st, "", subs)
        for i, s in enumerate(subs):
            want = np.sort(o.lookup_ids(rt, perm, st, "", int(s)))
            got = np.flatnonzero(np.unpackbits(bms[i].view(np.uint8), bitorder="little")).astype(np.uint32)
            assert np.array_equal(got, want), (name, s, got.size, want.size)
            assert counts[i] == want.size


def test_edge_cases(aclgpu):
    """Empty batch, empty graph, ragged batch sizes around the chunk size, invalid interned items."""
    from aclgpu import workloads
    w = workloads.c1()
    with aclgpu.Engine(w.schema) as e:
        items = e.make_items("namespace", "view", w.res, "user", "", w.subj)
        p, er = e.check_bulk_ids(items[:0])
        assert p.size == 0
        p, er = e.check_bulk_ids(items)  # no relationships at all
        assert (p == 1).all() and (er == 0).all()
        assert e.lookup_ids("namespace", "view", "user", "", 3).size == 0
        w.load(e)
        o = orc.Oracle(w.schema)
        w.load(o)
        rng = np.random.default_rng(1)
        for n in (1, 63, 64, 65, 1023, 1024, 1025, 4097):
            res = rng.integers(0, 1000, size=n).astype(np.uint32)
            sub = rng.integers(0, 1000, size=n).astype(np.uint32)
            p, er = e.check_bulk_ids(e.make_items("namespace", "view", res, "user", "", sub))
            op, oe = o.check_bulk_ids("namespace", "view", res, "user", "", sub)
            assert np.array_equal(p, op) and np.array_equal(er, oe), n
        bad = e.make_items("namespace", "view", [1, 2, 3], "user", "", [1, 2, 3])
        bad["permission"][1] = 77
        bad["resource_type"][2] = 999
        p, er = e.check_bulk_ids(bad)
        assert er.tolist()[1:] == [aclgpu.ERR_FAILED_PRECONDITION] * 2 and p.tolist()[1:] == [0, 0]
        # ids beyond the dense id space have no relationships
        far = e.make_items("namespace", "view", [4_000_000_000], "user", "", [5])
        assert e.check_bulk_ids(far)[0].tolist() == [1]


def test_walk_overflow_falls_back_and_backs_off(aclgpu, monkeypatch):
    """A block of the single-launch walk that outgrows its private frontier region hands the batch to the level loop (same answers);
    after such an overflow large batches skip the walk for 2, 4, ... passes instead of paying for a failed walk every time."""
    from aclgpu import workloads
    w = workloads.c4(scale=0.05, batch=40000)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rt, perm, st = w.check
    op, oe = o.check_bulk_ids_mt(4, rt, perm, w.res, st, "", w.subj)
    monkeypatch.setenv("ACL_LOCAL_CAP", "256")  # (read at acl_open)
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        items = e.make_items(rt, perm, w.res, st, "", w.subj)
        e.stats_reset()
        for _ in range(7):
            p, er = e.check_bulk_ids(items)
            assert np.array_equal(p, op) and np.array_equal(er, oe)
        s_ = e.stats()
        assert s_["check_passes"] == 7 and s_["local_passes"] == 0 and s_["expand_launches"] > 0
        # small batches still try the walk (and fall back) every time: nothing to back off from at their cost
        p, er = e.check_bulk_ids(items[:64])
        assert np.array_equal(p, op[:64]) and np.array_equal(er, oe[:64])


def test_direct_task_lists_step_aside_on_wide_fanout(aclgpu):
    """Round 5: the deep levels' direct task lists map a pair of segments' children through a 2 048-item head-bit window.  Nested groups with 40
    subgroups each overrun it: that walk raises kOverflowDirect, the batch is redone on the level loop, and the NEXT walk builds its lists the
    general way -- same answers every time, and the walk is back in service at once (no back-off: nothing was wrong with the frontier)."""
    from aclgpu import workloads
    rng = np.random.default_rng(5)
    fan, n_user, n_pod = 40, 4000, 3000
    g1 = np.arange(1, 1 + fan, dtype=np.uint32)                       # g0 -> 40 groups
    g2 = np.arange(1 + fan, 1 + fan + fan * fan, dtype=np.uint32)     # each of them -> 40 more
    gg_r = np.concatenate([np.zeros(fan, np.uint32), np.repeat(g1, fan)])
    gg_s = np.concatenate([g1, g2])
    gu_r = rng.choice(g2, size=n_user).astype(np.uint32)              # every user sits in one leaf group
    gu_s = np.arange(n_user, dtype=np.uint32)
    pods = np.arange(n_pod, dtype=np.uint32)
    o = orc.Oracle(workloads.SCHEMA_C4)
    with aclgpu.Engine(workloads.SCHEMA_C4) as e:
        for t in (o, e):
            t.add_edges("group", "member", "group", "member", gg_r, gg_s)
            t.add_edges("group", "member", "user", "", gu_r, gu_s)
            t.add_edges("pod", "namespace", "namespace", "", pods, np.zeros(n_pod, np.uint32))
            t.add_edges("pod", "viewer", "group", "member", pods, np.zeros(n_pod, np.uint32))
            t.add_edges("pod", "creator", "user", "", pods[:8], np.arange(8, dtype=np.uint32))
        n = 100000  # (>= the wide walk's threshold: 16 waves per unit, segments of ONE slot from level 3 on)
        res = rng.integers(0, n_pod, size=n).astype(np.uint32)
        sub = rng.integers(0, n_user + 500, size=n).astype(np.uint32)  # some users are in no group: full misses walk every level
        op, oe = o.check_bulk_ids_mt(4, "pod", "view", res, "user", "", sub)
        items = e.make_items("pod", "view", res, "user", "", sub)
        e.stats_reset()
        for _ in range(3):
            p, er = e.check_bulk_ids(items)
            assert np.array_equal(p, op) and np.array_equal(er, oe)
        s_ = e.stats()
        assert s_["check_passes"] == 3 and s_["expand_launches"] > 0 and s_["local_passes"] == 2, s_  # first walk tripped, the next two ran


def test_frontier_overflow_grows(aclgpu, monkeypatch):
    """A frontier too small for the batch is grown and the pass redone -- same answers.  (The LEVEL LOOP's frontier: the single-launch walk,
    whose blocks may or may not fit their regions at this size, is switched off for the first engine.)"""
    from aclgpu import workloads
    w = workloads.c4(scale=0.02, batch=600000, n_user=20000)
    o = orc.Oracle(w.schema)
    w.load(o)
    monkeypatch.setenv("ACL_LOCAL_MAX", "0")  # (read at acl_open)
    # the smallest legal frontier: one static chunk per wave + 1 dynamic chunk -> the deep levels overflow it
    with aclgpu.Engine(w.schema, frontier_entries=4096) as e:
        w.load(e)
        p, er = e.check_bulk_ids(e.make_items("pod", "view", w.res, "user", "", w.subj))
        m = 20000  # the oracle checks a prefix; the whole batch went through the retried pass
        op, oe = o.check_bulk_ids("pod", "view", w.res[:m], "user", "", w.subj[:m])
        assert np.array_equal(p[:m], op) and np.array_equal(er[:m], oe)
        assert e.stats()["overflow_retries"] >= 1
    monkeypatch.delenv("ACL_LOCAL_MAX")
    with aclgpu.Engine(w.schema) as e2:  # same batch with the default frontier (and the single-launch walk): identical bytes
        w.load(e2)
        p2, er2 = e2.check_bulk_ids(e2.make_items("pod", "view", w.res, "user", "", w.subj))
        assert np.array_equal(p, p2) and np.array_equal(er, er2)


def test_sub_batched_passes(aclgpu):
    """A batch larger than max_sub_batch is answered in several device passes (each with its own epilogue): same bytes as
    one pass, for the host-buffer and the device-resident entry points."""
    import torch
    from aclgpu import workloads
    w = workloads.c4(scale=0.02, batch=50000, n_user=20000)
    o = orc.Oracle(w.schema)
    w.load(o)
    op, oe = o.check_bulk_ids("pod", "view", w.res, "user", "", w.subj)
    with aclgpu.Engine(w.schema, max_sub_batch=4096) as e:
        w.load(e)
        items = e.make_items("pod", "view", w.res, "user", "", w.subj)
        p, er = e.check_bulk_ids(items)
        assert np.array_equal(p, op) and np.array_equal(er, oe)
        assert e.stats()["check_passes"] == (items.size + 4095) // 4096
        d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
        d_perm = torch.zeros(items.size, dtype=torch.uint8, device="cuda")
        d_err = torch.zeros(items.size, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        e.check_bulk_ids_device(d_items.data_ptr(), items.size, d_perm.data_ptr(), d_err.data_ptr())
        e.sync()
        assert np.array_equal(d_perm.cpu().numpy(), op) and np.array_equal(d_err.cpu().numpy(), oe)
        # the PostFilter keep mask over a sub-batched check
        off = np.arange(0, items.size + 1, 5, dtype=np.uint32)
        keep = e.check_bulk_keep_ids(items[:off[-1]], off)
        want = (op[:off[-1]].reshape(-1, 5) == 2).all(axis=1)
        assert np.array_equal(keep.astype(bool), want)


def test_string_entry_points_agree(aclgpu):
    """acl_check_bulk (NUL-terminated fields) and acl_check_bulk_v ({pointer, length} fields, no NUL behind the bytes) against single checks and the
    oracle: named objects, unknown names, `...` and real subject relations, and items that carry an error (check.go:55) in the middle of the batch --
    they are answered in place (order preserved, check.go:54-57), not compacted away."""
    schema = """definition user {}
definition group { relation member: user | group#member }
definition namespace { relation viewer: user | group#member
 relation creator: user
 permission view = viewer + creator }
definition pod { relation namespace: namespace
 relation creator: user
 permission view = creator + namespace->view }"""
    rels = [f"pod:ns{i % 7}/p{i}#namespace@namespace:ns{i % 7}" for i in range(300)] + [f"pod:ns{i % 7}/p{i}#creator@user:u{i % 31}" for i in range(300)]
    rels += [f"namespace:ns{i}#viewer@group:g{i}#member" for i in range(7)] + [f"group:g{i}#member@user:u{i + 40}" for i in range(7)] + ["group:g0#member@group:g1#member"]
    o = orc.Oracle(schema)
    o.write([(orc.OP_TOUCH, r) for r in rels])
    rng = np.random.default_rng(5)
    qs = []
    for k in range(6000):
        i, u = int(rng.integers(0, 330)), int(rng.integers(0, 60))
        qs.append(("pod", f"ns{i % 7}/p{i}", "view", "user", f"u{u}", "" if k % 3 else "..."))
    qs[10] = ("pod", "ns0/p0", "view", "group", "g0", "member")       # a subject with a relation
    qs[11] = ("namespace", "ns1", "view", "group", "g1", "member")
    qs[12] = ("nosuchtype", "x", "view", "user", "u1", "")            # FAILED_PRECONDITION
    qs[13] = ("pod", "ns0/p0", "nosuchperm", "user", "u1", "")
    qs[14] = ("pod", "ns0/p0", "view", "nosuchtype", "u1", "")
    qs[15] = ("pod", "ns0/p0", "view", "user", "u1", "nosuchrel")
    qs[16] = ("pod", "same", "view", "pod", "same", "")               # unknown object that is its own subject
    qs[5000] = ("pod", "never-written", "view", "user", "never-seen", "")
    # requests the API's validation refuses (validate.hpp): a single check answers INVALID_ARGUMENT, a bulk request fails AS A WHOLE
    invalid = [("pod", "", "view", "user", "u1", ""), ("", "", "", "", "", ""),   # empty fields: pkg/proxy/options_test.go:101-102
               ("pod", "ns0/kube-root-ca.crt", "view", "user", "u1", ""), ("pod", "ns0/p0", "view", "user", "system:admin", ""), ("pod", "ns0/p0", "view", "user", "*", ""),
               ("No_Such", "x", "view", "user", "u1", ""), ("pod", "ns0/p0", "vw", "user", "u1", "")]
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        want = [o.check(*q) for q in qs]
        for idx in (10, 11, 12, 13, 14, 15, 16, 5000):
            assert e.check(*qs[idx]) == want[idx], (idx, qs[idx])
        for bad in invalid:
            assert e.check(*bad) == (0, aclgpu.ERR_INVALID_ARGUMENT), bad
            if all(bad[:5]):
                assert o.check(*bad) == (0, aclgpu.ERR_INVALID_ARGUMENT), bad
            # (single-thread and pooled interning; from 2 048 items on the batch goes to the device in parts while the rest is interned: the ill-formed item in the
            #  last part, in the first)
            for batch in ([bad], qs[:50] + [bad] + qs[50:100], qs[:5000] + [bad], qs[:1000] + [bad] + qs[1000:5000]):
                with pytest.raises(aclgpu.AclError) as ei:
                    e.check_bulk(batch)
                assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT, bad
        # ACL_FLAG_PER_ITEM_VALIDATION (the patterns are a restatement from memory): the ill-formed item fails ITS pair, the call is answered
        with aclgpu.Engine(schema, "\n".join(rels), per_item_validation=True) as e_lax:
            for bad in invalid[2:]:
                for batch in (qs[:50] + [bad] + qs[50:100], qs[:5000] + [bad]):
                    got = e_lax.check_bulk(batch)
                    pairs = list(zip(got[0], got[1]))
                    at = 50 if len(batch) == 101 else 5000
                    assert pairs[at] == (0, aclgpu.ERR_INVALID_ARGUMENT), bad
                    assert pairs[:at] == want[:at] and pairs[at + 1:] == want[at:len(batch) - 1], bad
        for form, prep, call in (("c strings", e.make_check_strings_named(qs), e.check_bulk_prepared), ("views", e.make_check_views(qs), e.check_bulk_views)):
            p, er = call(prep)
            assert list(zip(p.tolist(), er.tolist())) == want, form
            p2, er2 = call((prep[0], 100, prep[2]))  # a small batch takes the single-thread path
            assert list(zip(p2.tolist(), er2.tolist())) == want[:100], form
        assert {w_[1] for w_ in want} >= {0, aclgpu.ERR_FAILED_PRECONDITION} and 0 < sum(w_[0] == 2 for w_ in want) < len(want)
        # the proxy's batches repeat themselves (one user for every pair of a PostFilter call, postfilter.go:88-119; F templates per list item): a
        # name equal to the previous item's is not looked up again (engine.cpp intern_items) -- runs of one subject, runs of one resource, the same
        # STRING under two types next to each other, unknown names and erroring items inside the runs, runs across the 16-item groups
        rep = []
        for k in range(5200):
            i = (k // 3) % 330
            u = f"u{(k // 37) % 60}" if (k // 500) % 2 == 0 else "never-seen"
            rep.append(("pod", f"ns{i % 7}/p{i}", "view", "user", u, ""))
        rep[40] = ("pod", "ns0/p0", "view", "group", "g0", "member")
        rep[41] = ("group", "g0", "member", "group", "g0", "member")   # the same name as resource and as subject, and as the item before it under another role
        rep[42] = ("group", "g0", "member", "user", "g0", "")          # ... and as a USER name: another table, must not take the group's id
        rep[43] = ("pod", "ns0/p0", "nosuchperm", "user", "g0", "")    # an erroring item inside a run
        rep[44] = ("pod", "ns0/p0", "view", "user", "g0", "")
        want_rep = [o.check(*q) for q in rep]
        for form, prep, call in (("c strings", e.make_check_strings_named(rep), e.check_bulk_prepared), ("views", e.make_check_views(rep), e.check_bulk_views)):
            p, er = call(prep)
            assert list(zip(p.tolist(), er.tolist())) == want_rep, form
        # acl_resolve_bulk_v: the same names as 16-byte items, no device pass; the id entry point then answers as the string one did
        for batch, want_b in ((rep, want_rep), (qs[:100], want[:100])):  # pooled and single-thread interning
            items, rerr = e.resolve_bulk_views(e.make_check_views(batch))
            p, er = e.check_bulk_ids(items)
            for k, (wp, we) in enumerate(want_b):
                if rerr[k]:
                    assert (wp, we) == (0, int(rerr[k])) and p[k] == 0 and er[k] != 0, (k, batch[k])
                else:
                    assert (int(p[k]), int(er[k])) == (wp, we), (k, batch[k])
            assert rerr.any() and not rerr.all()
            p2, er2 = call((prep[0], 100, prep[2]))
            assert list(zip(p2.tolist(), er2.tolist())) == want_rep[:100], form


@pytest.mark.parametrize("split", ["1", "2", "4"])
def test_host_batches_as_concurrent_slices(split, aclgpu, monkeypatch):
    """A host batch beyond one launch goes as sub-passes that alternate between streams of their own, each with a frontier region of its own
    (engine.cpp check_pass_local_host; ACL_HOST_SPLIT streams, 1 = one after the other); a lone caller's 262 144-item batch is cut in two the same
    way.  1 200 000 items (3 slices over 1 / 2 / 3 lanes), 262 144 and 300 001 items: every answer equals the oracle's whatever the lanes."""
    from aclgpu import workloads
    monkeypatch.setenv("ACL_HOST_SPLIT", split)  # (read at acl_open)
    w = workloads.c4(scale=0.02, batch=1200000, n_user=20000)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    op, oe = o.check_bulk_ids_mt(16, "pod", "view", w.res, "user", "", w.subj)
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        items = e.make_items("pod", "view", w.res, "user", "", w.subj)
        for n in (1200000, 262144, 300001):
            p, er = e.check_bulk_ids(items[:n])
            assert np.array_equal(p, op[:n]) and np.array_equal(er, oe[:n]), (split, n)
        st = e.stats()
        assert st["local_passes"] == st["check_passes"] and st["overflow_retries"] == 0  # the single-launch walk took every slice
        lanes = int(split)
        assert st["check_passes"] >= 4 and (lanes == 1 or st["check_passes"] > 4)  # (>= 2 slices for 1.2 M items whatever the lanes; more launches once batches are cut per lane)
