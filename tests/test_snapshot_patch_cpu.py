"""CPU tests of the write path's snapshot maintenance: committed writes are patched into the
HBM snapshot's host copy in place (plan.cpp patch_forward) and the result must describe exactly the store's live
relationships (verify_snapshot: every relationship findable by the kernels' search, nothing dead left, rows sorted,
no unsound leaf flag).  Runs on a store-only engine -- the device upload of the patched regions is covered by
tests/test_write_path_gpu.py."""
import random

import pytest
from hypothesis import given, settings, strategies as st

from tests.test_oracle_cross import SCHEMA, tuples_strategy
from tests.test_sharded_gloo import random_tuples


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


def writes_strategy():
    return st.lists(st.tuples(st.sampled_from([1, 2, 3]), tuples_strategy().map(lambda ts: ts[:6])), min_size=1, max_size=8)


def test_patch_after_every_write_matches_store(aclgpu):
    @settings(max_examples=60, deadline=None)
    @given(tuples_strategy(), writes_strategy())
    def run(initial, writes):
        e = aclgpu.Engine(SCHEMA, store_only=True)
        initial = list(dict.fromkeys(initial))
        if initial:
            e.write([(aclgpu.OP_TOUCH, t) for t in initial])
        assert e.selfcheck_snapshot() is False  # first call builds
        for op, ts in writes:
            ts = list(dict.fromkeys(ts))
            if not ts:
                continue
            try:
                e.write([(op, t) for t in ts])
            except aclgpu.AclError as ex:  # CREATE of an existing relationship: the whole write is refused
                assert ex.code == aclgpu.ERR_ALREADY_EXISTS
            e.selfcheck_snapshot()  # patch or rebuild, then verified inside the library (raises on any mismatch)
        e.close()

    run()


def test_patches_are_actually_used_and_leaf_flags_stay_sound(aclgpu):
    rng = random.Random(7)
    e = aclgpu.Engine(SCHEMA, store_only=True)
    e.write([(aclgpu.OP_TOUCH, t) for t in random_tuples(rng, 40)])
    e.selfcheck_snapshot()
    patched = 0
    for i in range(200):
        ts = random_tuples(rng, rng.randint(1, 5))
        op = rng.choice([aclgpu.OP_TOUCH, aclgpu.OP_TOUCH, aclgpu.OP_DELETE])
        e.write([(op, t) for t in ts])
        if i % 3 == 0:  # several writes may accumulate before the next read
            patched += e.selfcheck_snapshot()
    assert patched > 30, patched
    # a group that had no members gains a nested group: everything that flagged it as a leaf must be distrusted or fixed
    e2 = aclgpu.Engine(SCHEMA, store_only=True)
    e2.write([(aclgpu.OP_TOUCH, "group:g1#member@user:u1"), (aclgpu.OP_TOUCH, "group:g0#member@group:g1#member"),
              (aclgpu.OP_TOUCH, "doc:d0#viewer@group:g0#member"), (aclgpu.OP_TOUCH, "group:g2#member@user:u2")])
    e2.selfcheck_snapshot()
    e2.write([(aclgpu.OP_TOUCH, "group:g1#member@group:g2#member")])  # g1 was a leaf (only user members) until now
    assert e2.selfcheck_snapshot() is True
    e2.write([(aclgpu.OP_DELETE, "group:g1#member@group:g2#member"), (aclgpu.OP_DELETE, "group:g0#member@group:g1#member")])
    assert e2.selfcheck_snapshot() is True
    for e_ in (e, e2):
        e_.close()


def test_what_cannot_be_patched_is_rebuilt(aclgpu):
    e = aclgpu.Engine(SCHEMA, store_only=True)
    e.write([(aclgpu.OP_TOUCH, "doc:d0#viewer@user:u0")])
    e.selfcheck_snapshot()
    # first relationship ever in a class: every DECLARED class has descriptors and program ops from the build on, so this is a patch
    # (it used to rebuild the whole snapshot: the dual-write's lock tuple hits an empty class on every quiet proxy, workflow.go:392-418)
    e.write([(aclgpu.OP_TOUCH, "org:o0#admin@user:u0")])
    assert e.selfcheck_snapshot() is True
    e.write([(aclgpu.OP_TOUCH, "org:o1#admin@user:u1")])
    assert e.selfcheck_snapshot() is True
    # bulk loads bypass the change feed -> rebuild
    import numpy as np
    e.add_edges("doc", "viewer", "user", "", np.array([0], dtype=np.uint32), np.array([1], dtype=np.uint32))
    assert e.selfcheck_snapshot() is False
    # more new objects than the tables' headroom (16 384 + 25 %) -> rebuild, and the rebuilt tables fit again
    for base in range(0, 18000, 500):
        e.write([(aclgpu.OP_TOUCH, f"doc:new{base + i}#viewer@user:u0") for i in range(500)])
    assert e.selfcheck_snapshot() is False
    e.write([(aclgpu.OP_TOUCH, "doc:one-more#viewer@user:u0")])
    assert e.selfcheck_snapshot() is True
    e.close()


def test_expiring_relationships_and_the_patch_window(aclgpu):
    from tests import kat_runner
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], store_only=True)
    e.set_now(1000)
    e.write([(aclgpu.OP_TOUCH, ("workflow", "w1", "idempotency_key", "activity", "a1", ""), 1500)])
    e.selfcheck_snapshot()
    e.write([(aclgpu.OP_TOUCH, ("workflow", "w2", "idempotency_key", "activity", "a2", ""), 1200)])
    assert e.selfcheck_snapshot() is True  # an expiring relationship can be patched in; the window shrinks to 1200
    e.set_now(1300)  # a2 ran out: the relationship is patched OUT (an idempotency key expiring must not rebuild the graph)
    assert e.selfcheck_snapshot() is True
    assert [r[1] for r in e.read(rtype="workflow")] == ["w1"]
    e.set_now(1100)  # a clock set back (test clocks only): a2 is alive again, patched back in
    assert e.selfcheck_snapshot() is True
    assert sorted(r[1] for r in e.read(rtype="workflow")) == ["w1", "w2"]
    e.set_now(2000)  # both gone
    assert e.selfcheck_snapshot() is True
    assert e.read(rtype="workflow") == []
    e.close()


def test_expired_relationships_are_collected_after_the_gc_window(aclgpu):
    """Idempotency keys (activity.go:81-102) pile up at two per kube write; the store drops them 24 h after they expired (the reference engine's
    GC window, pkg/spicedb/spicedb.go:66).  A snapshot that slept through both the expiry and the collection still patches the rows out, and
    Watch never reports the collection (it is not an API write)."""
    from tests import kat_runner
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], store_only=True)
    t0 = 1_000_000
    e.set_now(t0)
    e.write([(aclgpu.OP_TOUCH, ("workflow", f"w{i}", "idempotency_key", "activity", f"a{i}", ""), t0 + 10 + i) for i in range(50)])
    e.write([(aclgpu.OP_TOUCH, ("namespace", "n0", "viewer", "user", "u0", ""))])
    e.selfcheck_snapshot()  # the snapshot holds all 50 keys
    _, cursor = e.watch_poll(aclgpu.WATCH_FROM_NOW)
    e.set_now(t0 + 24 * 3600 + 30)  # keys 0..20 expired at least 24 h ago, the rest less
    e.write([(aclgpu.OP_TOUCH, ("namespace", "n1", "viewer", "user", "u1", ""))])  # any write collects
    assert e.read(rtype="workflow") == []  # (all 50 have been invisible since their expiry)
    assert e.selfcheck_snapshot() is True  # patched: expiry crossings for the 29 still stored, feed entries for the 21 collected; verified inside
    events, cursor = e.watch_poll(cursor)
    assert [(ev[1], ev[2][0]) for ev in events] == [(aclgpu.OP_TOUCH, "namespace")], events
    e.set_now(t0 + 20)  # a test clock set back: what was collected stays gone, what was merely expired comes back
    assert sorted(r[1] for r in e.read(rtype="workflow")) == sorted(f"w{i}" for i in range(21, 50))  # (at <= now - 24 h: keys 0..20 were collected)
    assert e.selfcheck_snapshot() is True
    e.close()


def test_expiry_crossings_mixed_with_writes_and_compaction(aclgpu):
    """Random walk of the clock over many expiring idempotency keys, interleaved with writes, deletes and a background compaction's two
    halves: after every step the patched snapshot (forward and reverse rows) equals the store at that instant (acl_selfcheck_*)."""
    from tests import kat_runner
    b = kat_runner.load_bootstrap()
    rng = random.Random(23)
    e = aclgpu.Engine(b["schema"], store_only=True)
    now = 10_000
    e.set_now(now)
    e.write([(aclgpu.OP_TOUCH, ("workflow", "w-seed", "idempotency_key", "activity", "a-seed", ""), now + 50)])
    e.selfcheck_snapshot()
    patched = rebuilt = 0
    for step in range(160):
        k = rng.random()
        if k < 0.45:
            ups = []
            for i in rng.sample(range(41), rng.randint(1, 3)):  # (one update per relationship in a request)
                ups.append((rng.choice([aclgpu.OP_TOUCH, aclgpu.OP_TOUCH, aclgpu.OP_DELETE]), ("workflow", f"w{i}", "idempotency_key", "activity", f"a{i % 7}", ""),
                            now + rng.randint(1, 400)))
            e.write([(op, t, exp) if op != aclgpu.OP_DELETE else (op, t) for op, t, exp in ups])
        elif k < 0.55:
            e.write([(aclgpu.OP_TOUCH, ("namespace", f"n{rng.randint(0, 9)}", "viewer", "user", f"u{rng.randint(0, 9)}", ""))])
        elif k < 0.9:
            now += rng.choice([1, 5, 30, 120, 400])
            e.set_now(now)
        else:
            now -= rng.choice([1, 20, 200])  # (test clocks may run backwards)
            e.set_now(now)
        if step % 17 == 5:
            e.selfcheck_compaction(0)
            now += 60
            e.set_now(now)
            e.write([(aclgpu.OP_TOUCH, ("workflow", f"wc{step}", "idempotency_key", "activity", "ac", ""), now + 10)])
            e.selfcheck_compaction(1)
        code = e.selfcheck_snapshot_code()
        patched += code == 1
        rebuilt += code == 0
    assert patched > 60 and rebuilt == 0, (patched, rebuilt)  # (not even a class coming alive for the first time rebuilds)
    e.close()


@pytest.mark.parametrize("world", [2, 3])
def test_patching_a_shard(aclgpu, world):
    """A shard only patches rows of the types it owns; its snapshot must match the store restricted to those types."""
    rng = random.Random(11)
    engines = []
    for r in range(world):
        e = aclgpu.Engine(SCHEMA, store_only=True)
        e._check(e._L.acl_shard_configure(e._h, r, world))
        engines.append(e)
    init = random_tuples(rng, 40)
    for e in engines:
        e.write([(aclgpu.OP_TOUCH, t) for t in init])
        e.selfcheck_snapshot()
    for _ in range(40):
        ts = random_tuples(rng, 3)
        op = rng.choice([aclgpu.OP_TOUCH, aclgpu.OP_DELETE])
        for e in engines:
            e.write([(op, t) for t in ts])
            e.selfcheck_snapshot()
    for e in engines:
        e.close()


def test_background_compaction_host_half(aclgpu):
    """Snapshot compaction (engine.cpp): a snapshot built from a COPY-ON-WRITE view of the store while writes keep landing,
    then caught up with the ordinary patcher and adopted, must describe exactly the store -- including writes that hit the very
    tables the build was reading (they clone the table first) and deletes of relationships the view still held."""
    rng = random.Random(11)
    e = aclgpu.Engine(SCHEMA, store_only=True)
    e.write([(aclgpu.OP_TOUCH, t) for t in random_tuples(rng, 60)])
    e.selfcheck_snapshot()
    adopted = 0
    for round_ in range(30):
        e.selfcheck_compaction(0)  # the worker's part: view + build
        for _ in range(rng.randint(0, 12)):  # writes racing with the "build": they must not disturb the view and must all be caught up
            ts = random_tuples(rng, rng.randint(1, 6))
            e.write([(rng.choice([aclgpu.OP_TOUCH, aclgpu.OP_TOUCH, aclgpu.OP_DELETE]), t) for t in ts])
            if rng.random() < 0.3:
                e.selfcheck_snapshot()  # reads in between keep patching the OLD snapshot
        adopted += e.selfcheck_compaction(1)  # the adopting reader's part: catch up, swap, verify (raises on any mismatch)
        e.selfcheck_snapshot()
    assert adopted >= 25, adopted
    # a bulk load between the phases is not in the change feed: the build must be discarded, not adopted
    e.selfcheck_compaction(0)
    e.add_edges("doc", "creator", "user", "", [0], [0])
    assert e.selfcheck_compaction(1) is False
    e.selfcheck_snapshot()
    e.close()


def test_view_is_isolated_from_writes(aclgpu):
    """Copy-on-write tables: after the view is taken, 2 000 inserts and deletes in the live store leave a later adoption
    exact (the view's rows were never touched) -- on a table big enough that an in-place edit would shift thousands of keys."""
    rng = random.Random(3)
    e = aclgpu.Engine(SCHEMA, store_only=True)
    big = [("doc", f"d{i}", "viewer", "user", f"u{i % 97}", "") for i in range(4000)]
    for i in range(0, len(big), 1000):
        e.write([(aclgpu.OP_TOUCH, t) for t in big[i:i + 1000]])
    e.selfcheck_snapshot()
    e.selfcheck_compaction(0)
    for i in range(0, 2000, 500):
        e.write([(aclgpu.OP_DELETE, t) for t in big[i:i + 500]])
        e.write([(aclgpu.OP_TOUCH, ("doc", f"n{i + k}", "viewer", "user", f"u{k % 13}", "")) for k in range(500)])
    assert e.selfcheck_compaction(1) is True
    assert len(e.read(rtype="doc", rel="viewer")) == 4000
    e.close()


def test_long_random_write_streams_keep_the_snapshot_exact(aclgpu):
    """tools/fuzz_gpu.py --patcher: 1 500 write batches / filter deletes on the nested-group schema (and 300 large bursts on a wider universe), the
    host snapshot updated and verified against the store after every one."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    codes = fz.run_patcher(5, 1500)
    assert codes[1] > 1400  # patched in place, not rebuilt
    codes = fz.run_patcher(6, 300, universe=8, burst=400)
    assert codes[0] + codes[1] + codes[2] == 300 and codes["adopted"] >= 1 and codes["dropped"] == 0  # (background builds adopted with the writes since replayed)
    # the same stream on the schema with exclusions, intersections, wildcard classes and a non-monotone userset subject (fuzz_gpu.SCHEMA_COMBINE)
    codes = fz.run_patcher(7, 400, schema="combine")
    assert codes[1] > 380 and codes["dropped"] == 0


def test_object_ids_are_recycled_and_the_snapshot_stays_exact(aclgpu, monkeypatch):
    """an object that has lost its last relationship gives its id to the next NEW name of its type (after a quarantine:
    zero here), so the dense id spaces -- and every table sized by them -- stop growing under the dual-write stream (a new lock, workflow and
    two activities per kube write: workflow.go:392-462, activity.go:80-102).  The patched host snapshot is verified against the store all
    along; names resolve to the right ids before and after; ids handed out by acl_intern and ids still referenced are never taken; a Watch
    cursor from before an id changed its name is refused instead of replaying the old change under the new name."""
    from oracle import orc
    from tests import kat_runner
    monkeypatch.setenv("ACL_ID_QUARANTINE_MS", "0")  # (read when the schema is loaded)
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
    o = orc.Oracle(b["schema"])
    o.write([(orc.OP_TOUCH, r) for r in b["relationships"]])
    now = 1_700_000_000
    e.set_now(now)
    o.set_now(now)
    pinned = e.intern("lock", "kept-by-its-caller")
    assert e.selfcheck_snapshot() is False
    _u, first_cursor = e.watch_poll(aclgpu.WATCH_FROM_NOW)
    counts = []
    for i in range(600):
        lock = ("lock", f"h{i}", "workflow", "workflow", f"w{i}", "")
        w1 = [(aclgpu.OP_CREATE, ("pod", f"ns/p{i % 40}", "creator", "user", f"u{i % 7}", "")) if i < 40 else (aclgpu.OP_TOUCH, ("pod", f"ns/p{i % 40}", "viewer", "user", f"u{i % 7}", "")),
              (aclgpu.OP_CREATE, lock), (aclgpu.OP_CREATE, ("workflow", f"w{i}", "idempotency_key", "activity", f"a{i}", ""), now + 5)]
        w2 = [(aclgpu.OP_DELETE, lock), (aclgpu.OP_CREATE, ("workflow", f"w{i}", "idempotency_key", "activity", f"b{i}", ""), now + 5)]
        for ups in (w1, w2):
            e.write(ups, [(aclgpu.PRE_MUST_NOT_MATCH, dict(rtype="lock", rid=f"h{i}", rel="workflow", stype="workflow"))] if ups is w1 else ())
            o.write(ups)
            assert e.selfcheck_snapshot() is True, i  # patched in place, verified against the store
        now += 24 * 3600 // 50  # the clock runs: keys expire after 5 s and are collected 24 h later (spicedb.go:66) -- every 50 kube writes here
        e.set_now(now)
        o.set_now(now)
        if i % 25 == 0:
            for t in ("lock", "workflow", "pod"):
                assert sorted(e.read(rtype=t)) == sorted(o.read(rtype=t)), (i, t)
        counts.append((e.object_count("lock"), e.object_count("workflow"), e.object_count("activity")))
    # the id spaces plateau: locks are free again right after W2, workflows / activities once their keys are collected
    assert counts[-1][0] <= 3 and counts[-1][1] <= 60 and counts[-1][2] <= 120, counts[-1]
    assert counts[-1] == counts[300], (counts[300], counts[-1])
    assert e.stats()["ids_recycled"] > 1500
    assert e.find("lock", "kept-by-its-caller") == pinned and e.find("lock", "h5") is None and e.object_name("lock", pinned) == "kept-by-its-caller"
    # ids still referenced are never taken: every pod keeps its id and its relationships
    assert e.object_count("pod") == 40 and len(e.read(rtype="pod")) == len(o.read(rtype="pod"))
    # a cursor from before the ids changed their names cannot be replayed
    with pytest.raises(aclgpu.AclError) as ei:
        e.watch_poll(first_cursor, ["lock"])
    assert ei.value.code == aclgpu.ERR_OUT_OF_RANGE
    ups_now, _c = e.watch_poll(e.revision - 1)
    assert all(u[0] == e.revision for u in ups_now)
    # one request naming several NEW objects while ids are free, plus an object that is free right now (u-free lost its only relationship a
    # write ago): every name keeps an id of its own -- what the write names is held until its relationships are in
    e.write([(aclgpu.OP_TOUCH, ("pod", "ns/once", "viewer", "user", "u-free", ""))])
    e.write([(aclgpu.OP_DELETE, ("pod", "ns/once", "viewer", "user", "u-free", ""))])
    many = [(aclgpu.OP_TOUCH, ("pod", f"ns/fresh{k}", "viewer", "user", "u-free" if k == 3 else f"u-new{k}", "")) for k in range(6)]
    e.write(many)
    got = sorted(r[:6] for r in e.read(rtype="pod") if r[1].startswith("ns/fresh"))
    assert got == sorted(u[1] for u in many), got
    assert len({e.find("pod", f"ns/fresh{k}") for k in range(6)}) == 6 and len({e.find("user", f"u-new{k}") for k in (0, 1, 2, 4, 5)} | {e.find("user", "u-free")}) == 6
    # ... and a NEW subject named by several updates of one request keeps every one of its references
    e.write([(aclgpu.OP_TOUCH, ("pod", f"ns/fresh{k}", "creator", "user", "u-shared", "")) for k in range(5)])
    e.write([(aclgpu.OP_DELETE, ("pod", "ns/fresh3", "creator", "user", "u-shared", ""))])
    e.write([(aclgpu.OP_TOUCH, ("pod", "ns/fresh0", "viewer", "user", "u-latest", ""))])  # (must not be handed u-shared's id)
    assert sorted(r[1] for r in e.read(rtype="pod", stype="user", sid="u-shared")) == [f"ns/fresh{k}" for k in (0, 1, 2, 4)]
    assert e.find("user", "u-latest") != e.find("user", "u-shared")
    assert e.selfcheck_snapshot() is True
    e.close()


def test_an_id_handed_out_by_name_restarts_its_quarantine(aclgpu, monkeypatch):
    """the recycling quarantine of an unreferenced object counts from the last time its id was handed to a caller by name
    (acl_find, a resolved single Check, the subject of a LookupResources), not from when it became free -- otherwise a subject without
    relationships that a request has just resolved could be renamed by a concurrent write before the request is evaluated."""
    import time
    from tests import kat_runner
    monkeypatch.setenv("ACL_ID_QUARANTINE_MS", "300")  # (read when the schema is loaded)
    b = kat_runner.load_bootstrap()
    for handed_out in (False, True):
        e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
        e.write([(aclgpu.OP_TOUCH, ("pod", "ns/a", "viewer", "user", "u-x", ""))])
        e.write([(aclgpu.OP_DELETE, ("pod", "ns/a", "viewer", "user", "u-x", ""))])  # u-x takes part in no relationship any more
        ux = e.find("user", "u-x")
        time.sleep(0.4)  # ... and has sat out the quarantine
        if handed_out:
            assert e.find("user", "u-x") == ux  # a caller resolves the name: the id is theirs for another quarantine
        e.write([(aclgpu.OP_TOUCH, ("pod", "ns/b", "viewer", "user", "u-new", ""))])
        if handed_out:
            assert e.find("user", "u-x") == ux and e.find("user", "u-new") != ux
        else:
            assert e.find("user", "u-x") is None and e.find("user", "u-new") == ux  # (the control: without the hand-out the id is reused)
        e.close()


def test_expiry_maps_fold_and_views_stay_isolated(aclgpu):
    """Round 5: a class's expiry times live in an immutable sorted base + a small delta (ExpiryMap); Store::view() shares the base instead of
    copying one node per expiring key.  Thousands of expiring relationships (several folds), then updates, removals and re-creations on both
    sides of a fold: reads, the patched snapshot and a background build taken in the middle (phase 0 ... writes ... phase 1) all agree with
    the oracle, and keys expire exactly when their LATEST expiry says."""
    from oracle import orc
    from tests import kat_runner
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
    o = orc.Oracle(b["schema"])
    o.write([(orc.OP_TOUCH, r) for r in b["relationships"]])
    now = 1_700_000_000
    e.set_now(now)
    o.set_now(now)
    key = lambda j: ("workflow", f"w{j}", "idempotency_key", "activity", f"a{j}", "")  # noqa: E731
    N = 9000  # (> 2 x the delta's fold threshold)
    for i in range(0, N, 1000):
        ups = [(aclgpu.OP_CREATE, key(j), now + 100 + (j % 7)) for j in range(i, i + 1000)]
        e.write(ups)
        o.write(ups)
    assert e.selfcheck_snapshot() is False
    e.selfcheck_compaction(0)  # a background build starts from a view taken HERE
    # ... while writers go on: longer lives, shorter lives, removals, re-creations without an expiry
    ups = [(aclgpu.OP_TOUCH, key(j), now + 1000) for j in range(0, N, 3)] + [(aclgpu.OP_TOUCH, key(j), now + 50) for j in range(1, 2000, 3)]
    dels = [(aclgpu.OP_DELETE, key(j)) for j in range(2, 3000, 3)]
    for chunk in (ups[:1000], ups[1000:2000], ups[2000:3000], ups[3000:], dels):
        e.write(chunk)
        o.write(chunk)
    forever = [(aclgpu.OP_TOUCH, key(j)) for j in range(2, 300, 3)]
    e.write(forever)
    o.write(forever)
    assert e.selfcheck_compaction(1) is True  # the view's build catches up with all of that and is verified against the store
    for t in (now + 60, now + 104, now + 200, now + 2000):
        e.set_now(t)
        o.set_now(t)
        assert e.selfcheck_snapshot() is True  # (the expiry crossings are patched, not rebuilt)
        assert sorted(e.read(rtype="workflow")) == sorted(o.read(rtype="workflow")), t
    assert len(e.read(rtype="workflow")) == 100  # only the re-created ones never expire
    e.close()
