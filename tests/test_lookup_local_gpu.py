"""Single-launch LookupResources (k_rev_local) on the GPU: the reverse walk behind pkg/authz/lookups.go:49-83 in ONE launch per batch.
Every bitmap must equal the level loop's (k_rev_expand, ACL_REV_LOCAL=0) AND the oracle's id set; the kernel must really be the one
that ran (stats), pinned result buffers and staged ones must agree, and a lookup that outgrows its block must fall back."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


def ids_of(row):
    return np.flatnonzero(np.unpackbits(row.view(np.uint8), bitorder="little")).astype(np.uint32)


def test_nested_groups_equal_level_loop_and_oracle(aclgpu, monkeypatch):
    """C4's shape (5-level nested groups, arrows, userset subjects): non-terminal states are pushed through several reverse levels."""
    from aclgpu import workloads
    w = workloads.c4(scale=0.02, batch=1000, n_user=5000)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rng = np.random.default_rng(11)
    users = rng.integers(0, w.nobjects["user"], size=40).astype(np.uint32)
    groups = rng.integers(0, w.nobjects["group"], size=24).astype(np.uint32)
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        e.stats_reset()
        got = {}
        for rt, perm in (("pod", "view"), ("namespace", "view"), ("group", "member")):
            got[(rt, perm, "user")] = e.lookup_ids_batch(rt, perm, "user", "", users)
            got[(rt, perm, "group")] = e.lookup_ids_batch(rt, perm, "group", "member", groups)  # subject with a relation: it is its own member
        st = e.stats()
        assert st["rev_local_passes"] == 6 and st["expand_launches"] == 0 and st["lookup_requests"] == 3 * (40 + 24)
        assert st["levels_last"] >= 2
    monkeypatch.setenv("ACL_REV_LOCAL", "0")  # (read at acl_open)
    with aclgpu.Engine(w.schema) as e2:
        w.load(e2)
        e2.stats_reset()
        for (rt, perm, stype), (bms, counts) in got.items():
            subs, srel = (users, "") if stype == "user" else (groups, "member")
            b2, c2 = e2.lookup_ids_batch(rt, perm, stype, srel, subs)
            assert np.array_equal(bms, b2) and np.array_equal(counts, c2), (rt, perm, stype)
            for i in range(0, subs.size, 5):
                want = np.sort(o.lookup_ids(rt, perm, stype, srel, int(subs[i])))
                assert np.array_equal(ids_of(bms[i]), want), (rt, perm, stype, int(subs[i]))
                assert counts[i] == want.size
        s2 = e2.stats()
        assert s2["rev_local_passes"] == 0 and s2["expand_launches"] > 0
    assert max(int(c.max()) for _b, c in got.values()) > 50  # (the graph is not degenerate)
    # the result slot's rows in HBM (what a type of more than 1 M objects gets) instead of LDS: same rows
    monkeypatch.delenv("ACL_REV_LOCAL")
    monkeypatch.setenv("ACL_REV_LDS_ROWS", "0")
    with aclgpu.Engine(w.schema) as e3:
        w.load(e3)
        e3.stats_reset()
        for (rt, perm, stype), (bms, counts) in got.items():
            subs, srel = (users, "") if stype == "user" else (groups, "member")
            b3, c3 = e3.lookup_ids_batch(rt, perm, stype, srel, subs)
            assert np.array_equal(bms, b3) and np.array_equal(counts, c3), (rt, perm, stype)
        assert e3.stats()["rev_local_passes"] == 6
    # ... with every terminal row of the result slot DEFERRED to the chip-wide launch (round 6: what the heavy rounds of a lookup over a big type do;
    # ACL_REV_DEFER_MIN=1 makes every round of this small graph one), single lookups (the completion word raised by the third launch) and batches
    monkeypatch.setenv("ACL_REV_DEFER_MIN", "1")
    with aclgpu.Engine(w.schema) as e4:
        w.load(e4)
        e4.stats_reset()
        for (rt, perm, stype), (bms, counts) in got.items():
            subs, srel = (users, "") if stype == "user" else (groups, "member")
            b4, c4 = e4.lookup_ids_batch(rt, perm, stype, srel, subs)
            assert np.array_equal(bms, b4) and np.array_equal(counts, c4), (rt, perm, stype)
            for i in (0, subs.size // 2, subs.size - 1):
                b1, c1 = e4.lookup_ids_batch(rt, perm, stype, srel, [int(subs[i])])
                assert np.array_equal(b1[0], bms[i]) and c1[0] == counts[i], (rt, perm, stype, i)
            b4b, c4b = e4.lookup_ids_batch(rt, perm, stype, srel, subs)  # (the rows were handed back all zero: a second walk finds nothing stale)
            assert np.array_equal(bms, b4b) and np.array_equal(counts, c4b), (rt, perm, stype)
        assert e4.stats()["rev_local_passes"] >= 6 * 5 and e4.stats()["expand_launches"] == 0
    # ... and with the round-5 form (ONE block walks, copies and clears the row in HBM): the A/B knob's other side stays correct
    monkeypatch.delenv("ACL_REV_DEFER_MIN")
    monkeypatch.setenv("ACL_REV_BIG_ROWS", "0")
    with aclgpu.Engine(w.schema) as e5:
        w.load(e5)
        for (rt, perm, stype), (bms, counts) in got.items():
            subs, srel = (users, "") if stype == "user" else (groups, "member")
            b5, c5 = e5.lookup_ids_batch(rt, perm, stype, srel, subs)
            assert np.array_equal(bms, b5) and np.array_equal(counts, c5), (rt, perm, stype)


def test_pinned_and_pageable_result_rows_agree(aclgpu):
    from aclgpu import workloads
    w = workloads.c3(scale=0.1, batch=256, power_users=16)
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        rt, perm, st = w.check
        subs = np.concatenate([w.lookup_subjects, np.arange(5, dtype=np.uint32)])
        words = max(1, (e.object_count(rt) + 31) // 32)
        hb = e.host_alloc(subs.size * words * 4 + subs.size * 8)
        hb[:] = 0xFF
        pinned = (hb[:subs.size * words * 4].view(np.uint32).reshape(subs.size, words), hb[subs.size * words * 4:].view(np.uint64))
        b1, c1 = e.lookup_ids_batch(rt, perm, st, "", subs, out=pinned)
        assert b1 is pinned[0]
        b2, c2 = e.lookup_ids_batch(rt, perm, st, "", subs)
        assert np.array_equal(b1, b2) and np.array_equal(c1, c2)
        assert c1[:16].min() > 100 and int(c1.sum()) == int(sum(ids_of(r).size for r in b1))
        # a row wider than the type's id space (the caller's `words` may exceed what the snapshot covers): the tail is zeroed
        wide = np.full((3, words + 7), 0xFFFFFFFF, dtype=np.uint32)
        cnt = np.zeros(3, dtype=np.uint64)
        e._check(e._L.acl_lookup_resources_batch(e._h, e.type_id(rt), e.relation_id(rt, perm), e.type_id(st), e.relation_id(st, ""),
                                                 subs[:3].ctypes.data, 3, wide.ctypes.data, words + 7, cnt.ctypes.data))
        assert np.array_equal(wide[:, :words], b1[:3]) and not wide[:, words:].any() and np.array_equal(cnt, c1[:3])
        e.host_free(hb)


def test_depth_limit_and_cycles_reverse(aclgpu):
    """first visit wins at the level it is produced: a 60-long chain is cut at distance 50 exactly as the level loop and the oracle cut it;
    cyclic nesting terminates"""
    schema = "definition user {}\ndefinition group { relation member: user | group#member }"
    n = 60
    rels = [f"group:g{i}#member@group:g{i+1}#member" for i in range(n)] + [f"group:g{n}#member@user:deep"]
    rels += ["group:c0#member@group:c1#member", "group:c1#member@group:c2#member", "group:c2#member@group:c0#member", "group:c1#member@user:loop"]
    o = orc.Oracle(schema)
    o.write([(orc.OP_TOUCH, r) for r in rels])
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        e.stats_reset()
        deep = e.lookup("group", "member", "user", "deep")
        assert deep == o.lookup("group", "member", "user", "deep") and len(deep) == 50
        assert e.lookup("group", "member", "user", "loop") == o.lookup("group", "member", "user", "loop") == {"c0", "c1", "c2"}
        assert e.lookup("group", "member", "group", "g30", "member") == o.lookup("group", "member", "group", "g30", "member")
        assert e.stats()["rev_local_passes"] == 3


def test_block_that_outgrows_its_region_falls_back(aclgpu, monkeypatch):
    from aclgpu import workloads
    w = workloads.c3(scale=0.2, batch=256, power_users=8)
    # an ordinary user made a direct viewer of every namespace: 200 first-level states (namespace#viewer) and 200 behind them (namespace#view) in a log of 256.
    # (Round 6: 700 direct POD grants no longer do it -- a relation that only feeds the result permission is marked through, its states never enter the log.)
    big, nns = 7, w.nobjects["namespace"]
    assert nns >= 150
    w.edges.append(("namespace", "viewer", "user", "", np.arange(nns, dtype=np.uint32), np.full(nns, big, dtype=np.uint32)))
    w.edges.append(("pod", "viewer", "user", "", np.arange(100, 800, dtype=np.uint32), np.full(700, big, dtype=np.uint32)))
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    monkeypatch.setenv("ACL_LOCAL_CAP", "256")  # private frontier region of 256 entries per lookup (read at acl_open)
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        rt, perm, st = w.check
        subs = np.concatenate([w.lookup_subjects[:4], np.asarray([big], dtype=np.uint32)])
        e.stats_reset()
        bms, counts = e.lookup_ids_batch(rt, perm, st, "", subs)  # one lookup of the group overflows: the whole group takes the level loop
        s_ = e.stats()
        assert s_["overflow_retries"] >= 1 and s_["expand_launches"] > 0 and s_["rev_local_passes"] == 0
        for i, s in enumerate(subs):
            want = np.sort(o.lookup_ids(rt, perm, st, "", int(s)))
            assert np.array_equal(ids_of(bms[i]), want) and counts[i] == want.size
        assert counts[-1] >= 700
        # the others fit: the single launch answers
        e.stats_reset()
        b4, c4 = e.lookup_ids_batch(rt, perm, st, "", subs[:4])
        assert e.stats()["rev_local_passes"] == 1 and np.array_equal(b4, bms[:4]) and np.array_equal(c4, counts[:4])


@pytest.mark.parametrize("rows", ["lds", "hbm"])
def test_relation_that_only_feeds_the_result_permission_is_marked_through_exactly(aclgpu, monkeypatch, rows):
    """Round 6: a row whose children land in a relation whose only parent is the result permission (`doc#viewer` under `view = viewer + owner`) marks the
    permission's objects directly, one dispatch level further on -- so at the depth limit the shortcut must stop exactly where the two-step walk
    stopped: a doc whose viewer group sits K nested groups above the user is visible iff K + 2 <= 50 (reference pkg/spicedb/spicedb.go:34), in both
    row forms (result rows in the block's LDS; in HBM as bytes, with every row deferred to the chip-wide launch)."""
    schema = """
    definition user {}
    definition group { relation member: user | group#member }
    definition doc {
      relation viewer: user | group#member
      relation owner: user
      permission view = viewer + owner
    }
    """
    rels = [("group", "g1", "member", "user", "deep", "")] + [("group", f"g{i + 1}", "member", "group", f"g{i}", "member") for i in range(1, 55)]
    rels += [("doc", f"d{k}", "viewer", "group", f"g{k}", "member") for k in range(40, 56)] + [("doc", "direct", "viewer", "user", "deep", ""), ("doc", "own", "owner", "user", "deep", "")]
    o = orc.Oracle(schema)
    o.write([(orc.OP_TOUCH, r) for r in rels])
    want = o.lookup("doc", "view", "user", "deep")
    assert want == {f"d{k}" for k in range(40, 49)} | {"direct", "own"}  # (K + 2 <= 50)
    if rows == "hbm":
        monkeypatch.setenv("ACL_REV_LDS_ROWS", "0")
        monkeypatch.setenv("ACL_REV_DEFER_MIN", "1")
    with aclgpu.Engine(schema) as e:
        e.write([(aclgpu.OP_TOUCH, r) for r in rels])
        assert e.lookup("doc", "view", "user", "deep") == want
        assert e.lookup("doc", "viewer", "user", "deep") == o.lookup("doc", "viewer", "user", "deep") == {f"d{k}" for k in range(40, 50)} | {"direct"}  # the relation ITSELF asked for: K + 1 <= 50
        assert e.stats()["rev_local_passes"] == 2


def test_result_slots_that_other_permissions_expand_end_the_walk(aclgpu, monkeypatch):
    """A lookup's result slot that is a SINK of the reverse graph (Snapshot::rev_sink: nothing above it leads back into it) is marked and not expanded: `edit` under
    `view = viewer + edit + ...`, a relation that feeds permissions, namespace#admin under namespace#view under pod#view.  Rows in LDS, rows in HBM (the deferred
    chip-wide launches included: before this the walk of such a slot over a big type fell back to one block doing everything) -- every row against the oracle's ids
    and against the same engine with ACL_REV_SINK=0; group#member (it reaches itself through nested groups) keeps being expanded."""
    schema = """definition user {}
definition group { relation member: user | group#member }
definition namespace { relation owner: user | group#member
 relation viewer: user | group#member
 permission admin = owner
 permission view = viewer + admin }
definition pod { relation namespace: namespace
 relation editor: user | group#member
 relation creator: user
 relation viewer: user | group#member
 permission edit = editor + creator
 permission view = viewer + edit + namespace->view }"""
    rng = np.random.default_rng(5)
    nu, ng, nn, npod = 300, 40, 30, 4000
    rels = [f"group:g{g}#member@user:u{int(u)}" for g in range(ng) for u in rng.integers(0, nu, size=6)]
    rels += [f"group:g{g}#member@group:g{int(h)}#member" for g in range(ng) for h in rng.integers(g + 1, ng + 1, size=1) if h < ng]
    rels += [f"namespace:n{n}#owner@user:u{int(rng.integers(0, nu))}" for n in range(nn)] + [f"namespace:n{n}#viewer@group:g{int(rng.integers(0, ng))}#member" for n in range(nn)]
    for p in range(npod):
        rels.append(f"pod:p{p}#namespace@namespace:n{p % nn}")
        rels.append(f"pod:p{p}#creator@user:u{int(rng.integers(0, nu))}")
        if p % 3 == 0:
            rels.append(f"pod:p{p}#editor@group:g{int(rng.integers(0, ng))}#member")
        if p % 5 == 0:
            rels.append(f"pod:p{p}#viewer@user:u{int(rng.integers(0, nu))}")
    rels = list(dict.fromkeys(rels))
    o = orc.Oracle(schema)
    for b in range(0, len(rels), 1000):
        o.write([(orc.OP_TOUCH, r) for r in rels[b:b + 1000]])
    targets = [("pod", "edit"), ("pod", "creator"), ("pod", "editor"), ("pod", "view"), ("namespace", "admin"), ("namespace", "view"), ("group", "member")]
    users = [f"u{int(u)}" for u in rng.integers(0, nu, size=12)]
    want = {(rt, pm, u): sorted(o.lookup(rt, pm, "user", u)) for rt, pm in targets for u in users}
    assert sum(len(v) for v in want.values()) > 2000
    modes = [{}, {"ACL_REV_LDS_ROWS": "0"}, {"ACL_REV_LDS_ROWS": "0", "ACL_REV_DEFER_MIN": "1"}, {"ACL_REV_SINK": "0"}, {"ACL_REV_SINK": "0", "ACL_REV_LDS_ROWS": "0"},
             {"ACL_REV_LOCAL": "0"}, {"ACL_REV_LOCAL": "0", "ACL_REV_SINK": "0"}]  # (the level loop -- k_rev_expand -- skips the dead ops too)
    for env in modes:
        for k in ("ACL_REV_LDS_ROWS", "ACL_REV_DEFER_MIN", "ACL_REV_SINK", "ACL_REV_LOCAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)  # (read at acl_open)
        with aclgpu.Engine(schema, "\n".join(rels)) as e:
            e.stats_reset()
            for rt, pm in targets:
                sids = [e.intern("user", u) for u in users]
                bms, counts = e.lookup_ids_batch(rt, pm, "user", "", sids)
                for i, u in enumerate(users):
                    got = sorted(e.object_name(rt, int(x)) for x in ids_of(bms[i]))
                    assert got == want[(rt, pm, u)] and counts[i] == len(got), (env, rt, pm, u)
                one, c1 = e.lookup_ids_batch(rt, pm, "user", "", sids[:1])  # (a single lookup: the completion word)
                assert np.array_equal(one[0], bms[0]) and c1[0] == counts[0], (env, rt, pm)
            assert (e.stats()["expand_launches"] == 0) == ("ACL_REV_LOCAL" not in env)
