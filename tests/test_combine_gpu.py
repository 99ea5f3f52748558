"""GPU parity for schemas with intersection `&`, exclusion `-` and wildcard subjects `T:*` (the reference boots
arbitrary schemas: pkg/spicedb/spicedb.go:19-24, e2e/embedded_integration_test.go:34-250).  The kernels' combine instantiations
(k_check_local / k_expand <..., CMB>, the resolve pass, LookupResources = candidates + forward Check) against BOTH oracle restatements:
bit-exact (permissionship, error) per item, identical id sets per lookup."""
import numpy as np
import pytest

from oracle import orc
from oracle.pyoracle import PyOracle
from tests.outcomes import outcome
from tests.test_oracle_cross import NM_QUERIES, PY2C, SCHEMA_NM, nm_tuples_strategy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


def test_combine_hypothesis(aclgpu):
    """Random small graphs (cyclic group nesting, wildcards in positive and subtracted operands, arrows into permissions that are themselves
    non-monotone, intersection arrows `.all()` over docs with several folders and folders with several parents -- also nested and under `-` / `&`
    -- depth errors on both sides of `&` / `-`): every answer equals the C oracle's AND the Python oracle's."""
    from hypothesis import given, settings

    eng = aclgpu.Engine(SCHEMA_NM)
    subjects = [("user", "u0", ""), ("user", "u3", ""), ("group", "g0", "member")]

    @settings(max_examples=60, deadline=None)
    @given(nm_tuples_strategy())
    def run(tuples):
        eng.load_bootstrap(SCHEMA_NM)
        co, po = orc.Oracle(SCHEMA_NM), PyOracle(SCHEMA_NM)
        tuples = list(dict.fromkeys(tuples))
        if tuples:
            co.write([(orc.OP_TOUCH, t) for t in tuples])
            eng.write([(aclgpu.OP_TOUCH, t) for t in tuples])
        for t in tuples:
            po.touch(*t)
        perms, errs = eng.check_bulk(NM_QUERIES)
        want = [co.check(*q) for q in NM_QUERIES]
        assert list(zip(perms, errs)) == want
        assert want == [PY2C[po.check(*q)] for q in NM_QUERIES]
        for s in subjects:
            for rt, p in [("doc", "view"), ("doc", "odd"), ("doc", "strict"), ("doc", "edit"), ("folder", "audit"), ("folder", "view"), ("group", "active"), ("doc", "nothing"),
                          ("doc", "everywhere"), ("doc", "vetted"), ("doc", "deep_all"), ("folder", "sealed")]:
                assert outcome(eng.lookup, rt, p, *s) == outcome(co.lookup, rt, p, *s), (rt, p, s)

    run()
    eng.close()


def test_precedence_and_three_valued_rules_on_the_gpu(aclgpu):
    """tests/test_oracle_cross.py::test_precedence_and_three_valued_rules through the engine: `-` loosest, then `&`, then `+`; an error in the
    subtracted operand only matters when the base holds; NO beats an error under `&`."""
    schema = """
    definition user {}
    definition g { relation member: user | g#member }
    definition d {
      relation a: user | g#member
      relation b: user | g#member
      relation c: user | g#member
      permission p1 = a + b - c
      permission p2 = a - b + c
      permission p3 = a & b + c
      permission p4 = a - b & c
      permission p5 = a - b - c
      permission deny_err = a - b
      permission and_err = a & b
      permission nested = (a - deny_err) + (and_err & c)
    }
    """
    rels = [("d", "x", "a", "user", "u", ""), ("d", "x", "c", "user", "u", ""), ("d", "y", "a", "user", "u", ""), ("d", "y", "b", "user", "u", ""),
            ("g", "g0", "member", "user", "deep", "")] + [("g", f"g{i + 1}", "member", "g", f"g{i}", "member") for i in range(60)] + [
            ("d", "e1", "a", "user", "deep", ""), ("d", "e1", "b", "g", "g59", "member"), ("d", "e2", "b", "g", "g59", "member"),
            ("d", "e3", "a", "g", "g59", "member"), ("d", "e3", "b", "user", "deep", ""), ("d", "e1", "c", "user", "deep", "")]
    co = orc.Oracle(schema)
    co.write([(orc.OP_TOUCH, r) for r in rels])
    qs = [("d", o, p, "user", u, "") for o in ("x", "y", "e1", "e2", "e3") for p in ("p1", "p2", "p3", "p4", "p5", "deny_err", "and_err", "nested") for u in ("u", "deep", "nobody")]
    with aclgpu.Engine(schema) as e:
        e.write([(aclgpu.OP_TOUCH, r) for r in rels])
        perms, errs = e.check_bulk(qs)
        assert list(zip(perms, errs)) == [co.check(*q) for q in qs]
        assert e.check("d", "e1", "deny_err", "user", "deep") == (0, aclgpu.ERR_DEPTH)
        assert e.check("d", "e2", "deny_err", "user", "deep") == (1, 0)
        assert e.check("d", "x", "p1", "user", "u") == (1, 0) and e.check("d", "x", "p3", "user", "u") == (2, 0)
        for p in ("p1", "p4", "deny_err", "nested"):
            for u in ("u", "deep"):
                assert outcome(e.lookup, "d", p, "user", u) == outcome(co.lookup, "d", p, "user", u), (p, u)
        # a depth-tainted lookup fails the CALL with the candidate's code (reference pkg/authz/lookups.go:75-83: the stream ends at the
        # first Recv error, responsefilterer.go:196-204 fails the list request): e1 is a candidate of `deep` and its Check errs
        for p in ("deny_err", "and_err"):
            with pytest.raises(aclgpu.AclError) as ei:
                e.lookup("d", p, "user", "deep")
            assert ei.value.code == aclgpu.ERR_DEPTH and "max depth exceeded" in str(ei.value), (p, str(ei.value))
        bm, cnt = None, None
        with pytest.raises(aclgpu.AclError) as ei:  # ... through the id-level batch entry point too, whichever lookup of the batch it is
            e.lookup_ids_batch("d", "deny_err", "user", "", [e.find("user", "u"), e.find("user", "deep")])
        assert ei.value.code == aclgpu.ERR_DEPTH
        assert e.lookup("d", "p3", "user", "u") == co.lookup("d", "p3", "user", "u") == {"x", "y"}  # (an untainted lookup of the same schema still answers)
    # ACL_FLAG_LENIENT_LOOKUP: the erring candidates are dropped and the call succeeds (rounds 4-5; the oracle's lenient form agrees)
    co.set_lenient_lookup(True)
    with aclgpu.Engine(schema, lenient_lookup=True) as e:
        e.write([(aclgpu.OP_TOUCH, r) for r in rels])
        for p in ("p1", "p4", "deny_err", "and_err", "nested"):
            for u in ("u", "deep"):
                assert e.lookup("d", p, "user", u) == co.lookup("d", p, "user", u), (p, u)


SCHEMA_BANS = """
definition user {}
definition group {
  relation member: user | group#member
  relation banned: user
  permission active = member - banned
}
definition namespace {
  relation viewer: user | group#member
  relation banned: user | user:*
  permission view = viewer - banned
}
definition pod {
  relation namespace: namespace
  relation viewer: user | group#active | user:*
  relation creator: user
  relation banned: user | group#member
  permission view = (viewer + creator + namespace->view) - banned
  permission strict = creator & namespace->view
  permission loose = viewer + creator + namespace->view
}
"""


def bans_graph(seed, n_user=4000, n_group=600, n_ns=200, n_pod=30000):
    rng = np.random.default_rng(seed)
    E = []  # (rtype, rel, stype, srel, res[], subj[])
    u = lambda n: rng.integers(0, n_user, size=n).astype(np.uint32)  # noqa: E731
    g = lambda n: rng.integers(0, n_group, size=n).astype(np.uint32)  # noqa: E731
    E.append(("group", "member", "user", "", g(6 * n_group), u(6 * n_group)))
    lo = rng.integers(0, n_group // 2, size=n_group).astype(np.uint32)           # nesting: groups of the upper half contain groups of the lower half (acyclic)
    hi = (n_group // 2 + rng.integers(0, n_group - n_group // 2, size=n_group)).astype(np.uint32)
    E.append(("group", "member", "group", "member", hi, lo))
    E.append(("group", "banned", "user", "", g(n_group), u(n_group)))
    E.append(("namespace", "viewer", "user", "", rng.integers(0, n_ns, size=8 * n_ns).astype(np.uint32), u(8 * n_ns)))
    E.append(("namespace", "viewer", "group", "member", rng.integers(0, n_ns, size=3 * n_ns).astype(np.uint32), g(3 * n_ns)))
    E.append(("namespace", "banned", "user", "", rng.integers(0, n_ns, size=2 * n_ns).astype(np.uint32), u(2 * n_ns)))
    E.append(("namespace", "banned", "user", "*", rng.integers(0, n_ns, size=n_ns // 20).astype(np.uint32), np.zeros(n_ns // 20, dtype=np.uint32)))
    pods = np.arange(n_pod, dtype=np.uint32)
    pod_ns, pod_creator = rng.integers(0, n_ns, size=n_pod).astype(np.uint32), u(n_pod)
    E.append(("pod", "namespace", "namespace", "", pods, pod_ns))
    E.append(("pod", "creator", "user", "", pods, pod_creator))
    E.append(("pod", "banned", "user", "", pods[::7], pod_creator[::7]))            # every 7th creator is banned from their own pod
    E.append(("namespace", "viewer", "user", "", pod_ns[::5], pod_creator[::5]))     # every 5th creator also views the pod's namespace (`strict`)
    E.append(("pod", "viewer", "user", "", rng.integers(0, n_pod, size=3 * n_pod).astype(np.uint32), u(3 * n_pod)))
    E.append(("pod", "viewer", "group", "active", rng.integers(0, n_pod, size=n_pod).astype(np.uint32), g(n_pod)))
    E.append(("pod", "viewer", "user", "*", rng.integers(0, n_pod, size=n_pod // 50).astype(np.uint32), np.zeros(n_pod // 50, dtype=np.uint32)))
    E.append(("pod", "banned", "user", "", rng.integers(0, n_pod, size=n_pod).astype(np.uint32), u(n_pod)))
    E.append(("pod", "banned", "group", "member", rng.integers(0, n_pod, size=n_pod // 4).astype(np.uint32), g(n_pod // 4)))
    return E, dict(user=n_user, group=n_group, namespace=n_ns, pod=n_pod)


def load_numeric(target, E):
    for rt, rel, st, sr, res, subj in E:
        target.add_edges(rt, rel, st, sr, res, subj)


@pytest.mark.parametrize("mode", ["walk", "level-loop", "overflow", "tiny-units"])
def test_bans_workload(mode, aclgpu, monkeypatch):
    """A pod graph with bans, wildcards and an `active = member - banned` userset under nested groups, 120 000 checks per permission through the
    interned-id bulk call: the single-launch walk (both instantiations), the level loop alone, and the walk overflowing into the level loop --
    all equal to the oracle; LookupResources for a few users equals the definition."""
    E, n = bans_graph(11)
    co = orc.Oracle(SCHEMA_BANS)
    load_numeric(co, E)
    co.freeze()
    if mode == "level-loop":
        monkeypatch.setenv("ACL_LOCAL_MAX", "0")
    if mode == "overflow":
        monkeypatch.setenv("ACL_LOCAL_CAP", "256")
    rng = np.random.default_rng(3)
    B = 120000 if mode != "tiny-units" else 3000
    res = rng.integers(0, n["pod"], size=B).astype(np.uint32)
    sub = rng.integers(0, n["user"], size=B).astype(np.uint32)
    # a share of requests that hit: the pod's creator / one of its viewers asks
    creators = dict(zip(E[8][4].tolist(), E[8][5].tolist()))
    assert E[8][1] == "creator"
    for i in range(0, B, 3):
        sub[i] = creators[int(res[i])]
    with aclgpu.Engine(SCHEMA_BANS) as e:
        load_numeric(e, E)
        for perm in ("view", "strict", "loose"):
            p, er = e.check_bulk_ids(e.make_items("pod", perm, res, "user", "", sub))
            op, oe = co.check_bulk_ids_mt(8, "pod", perm, res, "user", "", sub)
            assert np.array_equal(p, op) and np.array_equal(er, oe), (mode, perm, int((p != op).sum()))
            assert 0 < int((p == 2).sum()) < B
        p, er = e.check_bulk_ids(e.make_items("group", "active", rng.integers(0, n["group"], size=5000), "user", "", rng.integers(0, n["user"], size=5000)))
        st = e.stats()
        if mode == "walk":
            assert st["local_passes"] >= 4 and st["expand_launches"] == 0
        if mode == "level-loop":
            assert st["local_passes"] == 0 and st["expand_launches"] > 0
        # Filter: candidates + forward Check
        subs = [int(x) for x in rng.integers(0, n["user"], size=5)] + [int(sub[0])]
        for perm in ("view", "strict"):
            bms, counts = e.lookup_ids_batch("pod", perm, "user", "", subs)
            for i, s in enumerate(subs):
                want = np.sort(co.lookup_ids("pod", perm, "user", "", s))
                got = np.flatnonzero(np.unpackbits(bms[i].view(np.uint8), bitorder="little")).astype(np.uint32)
                assert np.array_equal(got, want), (mode, perm, s, got.size, want.size)
                assert counts[i] == want.size


def test_monotone_schema_takes_the_monotone_kernels(aclgpu):
    """A union-only schema never pays for the combine machinery: no boolean programs in its snapshot (the launchers pick the CMB kernels by that)."""
    from aclgpu import workloads
    w = workloads.c2(scale=0.02, batch=2000)
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        e.check_bulk_ids(e.make_items("pod", "view", w.res, "user", "", w.subj))
        assert e.stats()["local_passes"] == 1


def test_branching_cycles_under_exclusion_and_intersection(aclgpu):
    """Group nesting with BRANCHING cycles (g0 -> {g0, g1}, g1 -> g0, ...) under `-` and `&`: without merging, the pending sub-checks double per
    dispatch level for 50 levels.  A pass that outgrows its frontier is redone with identical (cell, state, level) entries merged
    (k_dedup_cells); answers equal the oracle's, which memoises on the same key: found by the live-graph fuzz (tools/fuzz_gpu.py --schema
    combine), pinned here.  A cycle THROUGH a non-monotone permission (teams whose members are teams' `active` members) is walked too -- every
    visit of such a state is a combine node of its own -- as long as it does not branch; where it branches the nodes double per level and the
    call fails LOUDLY with RESOURCE_EXHAUSTED instead of answering (the reference's engine, run with every cache off as the proxy runs it,
    spicedb.go:45-47, would be dispatching 2^50 sub-checks there)."""
    schema = """
definition user {}
definition group {
  relation member: user | group#member
  relation banned: user
  relation vip: user | group#member
  permission active = member - banned
  permission inner = member & vip
}
definition team {
  relation member: user | team#active
  relation banned: user
  permission active = member - banned
}
"""
    rels = ["group:g0#member@group:g0#member", "group:g0#member@group:g1#member", "group:g1#member@group:g0#member",
            "group:g1#member@group:g2#member", "group:g2#member@group:g0#member", "group:g2#member@group:g1#member", "group:g2#member@user:deep",
            "group:g2#member@user:outcast", "group:g0#banned@user:outcast", "group:g1#vip@group:g2#member", "group:g0#vip@user:deep",
            # a cycle through team#active (non-monotone) that does not branch: t0 <- t1#active <- t0#active
            "team:t0#member@team:t1#active", "team:t1#member@team:t0#active", "team:t1#member@user:deep", "team:t1#member@user:outcast", "team:t0#banned@user:outcast",
            # ... and one that does: b0 <- {b0, b1}#active, b1 <- {b0, b1}#active
            "team:b0#member@team:b0#active", "team:b0#member@team:b1#active", "team:b1#member@team:b0#active", "team:b1#member@team:b1#active", "team:b1#member@user:deep"]
    o = orc.Oracle(schema)
    o.write([(orc.OP_TOUCH, r) for r in rels])
    qs = [("group", g, p, "user", u, "") for g in ("g0", "g1", "g2") for p in ("active", "inner", "member") for u in ("deep", "outcast", "nobody")]
    qs += [("team", t, "active", "user", u, "") for t in ("t0", "t1") for u in ("deep", "outcast", "nobody")]
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        for q in qs:
            assert e.check(*q[:5]) == o.check(*q), q
        perms, errs = e.check_bulk(qs * 40)
        assert list(zip(perms, errs)) == [o.check(*q) for q in qs] * 40
        assert e.stats()["overflow_retries"] >= 1  # (the merging pass was what answered)
        for rt, perm in (("group", "active"), ("group", "inner")):
            for u in ("deep", "outcast"):
                assert e.lookup(rt, perm, "user", u) == set(o.lookup(rt, perm, "user", u)), (rt, perm, u)
        assert e.check("team", "b0", "active", "user", "deep") == (2, 0)  # (answered HAS before the tree of nodes outgrows anything)
        with pytest.raises(aclgpu.AclError) as ei:
            e.check("team", "b0", "active", "user", "nobody")
        assert ei.value.code == aclgpu.ERR_RESOURCE_EXHAUSTED


@pytest.mark.parametrize("mode", ["walk", "level-loop"])
def test_intersection_arrows(mode, aclgpu, monkeypatch):
    """`a.all(b)` (row a12's last construct but caveats; EXTERNAL, unverified semantics restated in oracle/acl_oracle.c EX_ARROW_ALL): every
    subject of the tupleset is dispatched into a result cell of its own, member nodes fold the verdicts into the arrow's cell -- no subject: NO,
    one NO decides, else an error beats HAS.  Known outcomes (a doc in two folders, one of them banning the user; no folder at all; under `&`,
    `+` and `-`), a tupleset of 3 000 folders (cells and member nodes reserved window by window; one dissenting folder among them), a depth
    error on one member, LookupResources -- on the single-launch walk and on the level loop, against the C oracle."""
    schema = """
definition user {}
definition group { relation member: user | group#member }
definition folder {
  relation viewer: user | group#member
  relation banned: user
  permission view = viewer - banned
}
definition doc {
  relation parent: folder
  relation owner: user
  permission view_all = parent.all(view)
  permission view_any = parent.any(view)
  permission strict = owner & parent.all(view)
  permission lax = owner + parent.all(view)
  permission outside = owner - parent.all(view)
}
"""
    rels = [("folder", "f1", "viewer", "user", "a", ""), ("folder", "f2", "viewer", "user", "a", ""), ("folder", "f2", "viewer", "user", "b", ""), ("folder", "f2", "banned", "user", "b", ""),
            ("doc", "d1", "parent", "folder", "f1", ""), ("doc", "d1", "parent", "folder", "f2", ""), ("doc", "d2", "parent", "folder", "f2", ""),
            ("doc", "d1", "owner", "user", "a", ""), ("doc", "d3", "owner", "user", "b", ""), ("doc", "d2", "owner", "user", "b", "")]
    # a doc in 3 000 folders that all let `a` and `c` in -- but one of them bans `c`
    rels += [("folder", f"w{i}", "viewer", "user", u, "") for i in range(3000) for u in ("a", "c")] + [("doc", "wide", "parent", "folder", f"w{i}", "") for i in range(3000)]
    rels += [("folder", "w1717", "banned", "user", "c", "")]
    # a folder whose viewers sit at the end of a 60-long group chain: `deep` is beyond the depth limit there, `a` is a direct viewer elsewhere
    rels += [("group", "g0", "member", "user", "deep", "")] + [("group", f"g{i + 1}", "member", "group", f"g{i}", "member") for i in range(60)]
    rels += [("folder", "far", "viewer", "group", "g59", "member"), ("folder", "near", "viewer", "user", "deep", ""), ("doc", "e1", "parent", "folder", "far", ""),
             ("doc", "e1", "parent", "folder", "near", ""), ("doc", "e2", "parent", "folder", "far", ""), ("doc", "e2", "parent", "folder", "f1", "")]
    co = orc.Oracle(schema)
    for i in range(0, len(rels), 900):
        co.write([(orc.OP_TOUCH, r) for r in rels[i:i + 900]])
    if mode == "level-loop":
        monkeypatch.setenv("ACL_LOCAL_MAX", "0")
    docs = ["d1", "d2", "d3", "wide", "e1", "e2", "ghost"]
    qs = [("doc", d, p, "user", u, "") for d in docs for p in ("view_all", "view_any", "strict", "lax", "outside") for u in ("a", "b", "c", "deep", "nobody")]
    with aclgpu.Engine(schema) as e:
        for i in range(0, len(rels), 900):
            e.write([(aclgpu.OP_TOUCH, r) for r in rels[i:i + 900]])
        perms, errs = e.check_bulk(qs * 3)
        assert list(zip(perms, errs)) == [co.check(*q) for q in qs] * 3
        assert e.check("doc", "d1", "view_all", "user", "a") == (2, 0) and e.check("doc", "d1", "view_all", "user", "b") == (1, 0)  # f2 bans b
        assert e.check("doc", "d3", "view_all", "user", "b") == (1, 0) and e.check("doc", "d3", "lax", "user", "b") == (2, 0)      # no folder: NO; the owner still gets `lax`
        assert e.check("doc", "wide", "view_all", "user", "a") == (2, 0) and e.check("doc", "wide", "view_all", "user", "c") == (1, 0)  # one dissenting folder of 3 000
        assert e.check("doc", "e1", "view_all", "user", "deep") == (0, aclgpu.ERR_DEPTH)  # near: HAS, far: beyond the depth limit -> the error wins over HAS
        assert e.check("doc", "e2", "view_all", "user", "deep") == (1, 0)                  # f1 says NO: decides whatever `far` would have said
        for p in ("view_all", "strict", "lax", "outside"):
            for u in ("a", "b", "c"):
                assert outcome(e.lookup, "doc", p, "user", u) == outcome(co.lookup, "doc", p, "user", u), (p, u)
        st = e.stats()
        if mode == "walk":
            assert st["local_passes"] >= 1
        else:
            assert st["local_passes"] == 0 and st["expand_launches"] > 0
