"""GPU tests of the callers either side of the kernels: the PostFilter keep mask
(pkg/authz/postfilter.go:58-182), the PreFilter IsAllowed over the LookupResources bitmap (lookups.go:25-36),
the micro-batching front-end for concurrent single checks (check.go:76-94) and the Watch -> re-check loop
(watch.go:27-111).  Expected answers come from the CPU oracle."""
import threading
import time

import numpy as np
import pytest

from oracle import orc
from tests import kat_runner

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


def test_postfilter_keep_mask(aclgpu):
    """K list items x ragged PostFilter pairs -> keep mask, ids path (AND on the device) and strings path; equals the
    reference's rule applied to the oracle's per-pair answers."""
    from aclgpu import workloads
    w = workloads.c2(scale=0.05, batch=6000)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    rng = np.random.default_rng(5)
    K = 2000
    nper = rng.integers(0, 4, size=K)  # 0..3 pairs per list item (0 = no template resolved -> kept)
    off = np.concatenate([[0], np.cumsum(nper)]).astype(np.uint32)
    n = int(off[-1])
    res, subj = w.res[:n].copy(), w.subj[:n].copy()
    operm, oerr = o.check_bulk_ids("pod", "view", res, "user", "", subj)
    want = np.array([all(operm[j] == 2 and oerr[j] == 0 for j in range(off[i], off[i + 1])) for i in range(K)])
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        items = e.make_items("pod", "view", res, "user", "", subj)
        keep = e.check_bulk_keep_ids(items, off)
        assert np.array_equal(keep.astype(bool), want)
        assert 0 < want.sum() < K
        # an invalid pair (unknown permission index) is a pair error -> its item is dropped
        bad = items.copy()
        victim = int(np.flatnonzero((nper > 0) & want)[0])
        bad["permission"][off[victim]] = 99
        keep2 = e.check_bulk_keep_ids(bad, off)
        assert not keep2[victim] and np.array_equal(np.delete(keep2, victim), np.delete(keep, victim))
        with pytest.raises(aclgpu.AclError):
            e.check_bulk_keep_ids(items, [0, n + 1])


def test_postfilter_one_subject_reverse_route(aclgpu):
    """Round 6.  A PostFilter call names ONE subject for all its pairs (reference pkg/authz/postfilter.go:67-119): acl_check_bulk_keep_v
    and acl_check_bulk_keep_packed answer such a call by one reverse walk + bit tests.  Compared here, on named objects: both entry points (route taken: the stats
    say so), the forward path (acl_check_bulk_v + the AND of postfilter.go:144-178) and the oracle -- for a power user, an ordinary user, a user nobody knows; with
    unknown pod names in the list, ragged pair ranges, and the calls the route must NOT take (two subjects, a userset subject) or must fail exactly as the forward
    path fails (an id the API refuses: the message names the field).  acl_check_bulk_packed answers as acl_check_bulk_v."""
    import ctypes
    from aclgpu import workloads
    w = workloads.c3(scale=0.05, batch=256, power_users=4)
    o = orc.Oracle(w.schema)
    w.load(o)
    o.freeze()
    npod, nuser = w.nobjects["pod"], w.nobjects["user"]
    pod_ns = np.zeros(npod, dtype=np.int64)
    for ed in w.edges:
        if ed[0] == "pod" and ed[1] == "namespace":
            pod_ns[ed[4]] = ed[5]
    names = {"pod": [f"ns{int(pod_ns[i])}/pod-{i}" for i in range(npod)], "user": [f"user-{i}" for i in range(nuser)]}
    rng = np.random.default_rng(11)
    with aclgpu.Engine(w.schema) as e:
        out = ctypes.c_uint32()
        for t_, ns_ in names.items():
            tid = e.type_id(t_)
            for nm in ns_:
                e._check(e._L.acl_intern(e._h, tid, nm.encode(), ctypes.byref(out)))
        w.load(e)

        def forward_pairs(items):  # names -> ids, then the id entry point: the forward walk whatever the pairs look like (a string call of one user's pairs
            it16, rerr = e.resolve_bulk_views(e.make_check_views(items))  # takes the reverse walk itself on this schema: below)
            assert not rerr.any()
            return e.check_bulk_ids(it16)

        def forward(items, off):
            p, er = forward_pairs(items)
            return np.array([all(p[j] == 2 and er[j] == 0 for j in range(off[i], off[i + 1])) for i in range(len(off) - 1)])

        K = 3000
        routed = 0
        for uid, uname in [(int(w.lookup_subjects[0]), None), (3, None), (None, "ghost-user")]:
            uname = uname or names["user"][uid]
            pods = rng.integers(0, npod, size=K)
            pnames = [names["pod"][int(p_)] if k % 13 else f"nowhere/pod-{k}" for k, p_ in enumerate(pods)]  # every 13th name is unknown to the table
            items = [("pod", pn, "view", "user", uname, "") for pn in pnames]
            off = np.arange(K + 1, dtype=np.uint32)
            if uid is None:
                want = np.zeros(K, dtype=bool)
            else:
                op, oe = o.check_bulk_ids("pod", "view", pods.astype(np.uint32), "user", "", np.full(K, uid, dtype=np.uint32))
                want = np.array([(k % 13 != 0) and op[k] == 2 and oe[k] == 0 for k in range(K)])
            before = e.stats()["keep_route_calls"]
            kv = e.check_bulk_keep_views(e.make_check_views(items), off)
            kp = e.check_bulk_keep_packed(e.make_check_packed(items), off)
            kc = e.check_bulk_keep(items, off)  # (the NUL-terminated form: the same route)
            assert e.stats()["keep_route_calls"] == before + 3, uname  # all three calls took the reverse walk
            routed += 3
            assert np.array_equal(kv.astype(bool), want) and np.array_equal(kp.astype(bool), want) and np.array_equal(np.asarray(kc).astype(bool), want), uname
            assert np.array_equal(forward(items, off), want), uname
            # CheckBulkPermissions itself -- what the unpatched proxy sends for a list (postfilter.go:134) -- takes the walk too on this schema (no recursion: no
            # Check of pod#view can end at the depth limit): every pair's permissionship AND error as the forward walk gives them, by all three string forms
            fp, fe = forward_pairs(items)
            before = e.stats()["keep_route_calls"]
            for got in (e.check_bulk_views(e.make_check_views(items)), e.check_bulk_packed(e.make_check_packed(items)), e.check_bulk_prepared(e.make_check_strings_named(items))):
                assert np.array_equal(got[0], fp) and np.array_equal(got[1], fe) and not fe.any(), uname
            assert e.stats()["keep_route_calls"] == before + 3, uname
            routed += 3
            if uid == int(w.lookup_subjects[0]):
                assert 0.02 < want.mean() < 0.9  # (the power user sees a good part of the list, not all of it)
        # a user whose reach CHANGES between calls: the engine remembers per subject whether the last walk allowed few or many objects (a hint for the host's
        # pass: resolve names while the device walks, or not) -- the masks must be right with the hint stale in either direction
        uname = names["user"][5]
        pods = rng.integers(0, npod, size=K)
        items = [("pod", names["pod"][int(p_)], "view", "user", uname, "") for p_ in pods]
        off = np.arange(K + 1, dtype=np.uint32)
        prep = e.make_check_views(items)
        granted = sorted({int(p_) for p_ in pods})[:2400]
        grants = [("pod", names["pod"][g_], "viewer", "user", uname, "") for g_ in granted]
        kept = []
        for phase in ("few", "few again", "many after the grants (hint: few)", "many again", "few after the deletes (hint: many)"):
            if phase.startswith("many after"):
                for b_ in range(0, len(grants), 1000):
                    e.write([(aclgpu.OP_TOUCH, g_) for g_ in grants[b_:b_ + 1000]])
            if phase.startswith("few after"):
                for b_ in range(0, len(grants), 1000):
                    e.write([(aclgpu.OP_DELETE, g_) for g_ in grants[b_:b_ + 1000]])
            before = e.stats()["keep_route_calls"]
            kv = e.check_bulk_keep_views(prep, off).astype(bool)
            assert e.stats()["keep_route_calls"] == before + 1 and np.array_equal(kv, forward(items, off)), phase
            kept.append(int(kv.sum()))
        assert kept[0] == kept[1] < K // 2 < kept[2] == kept[3] and 0 < kept[4] <= kept[0], kept  # (the deletes also take the viewer grants the user had before)
        # ragged pair ranges (0..3 pairs per list item, all for one user and permission: an item without pairs is kept, postfilter.go:145-150)
        uname = names["user"][int(w.lookup_subjects[1])]
        nper = rng.integers(0, 4, size=1500)
        off = np.concatenate([[0], np.cumsum(nper)]).astype(np.uint32)
        items = [("pod", names["pod"][int(p_)], "view", "user", uname, "") for p_ in rng.integers(0, npod, size=int(off[-1]))]
        before = e.stats()["keep_route_calls"]
        kv = e.check_bulk_keep_views(e.make_check_views(items), off)
        assert e.stats()["keep_route_calls"] == before + 1 and np.array_equal(kv.astype(bool), forward(items, off)) and kv[nper == 0].all()
        # calls the route must not take: two subjects; a userset subject -- same masks as the forward path, the counter stands still
        two = [("pod", names["pod"][int(p_)], "view", "user", names["user"][int(w.lookup_subjects[k % 2])], "") for k, p_ in enumerate(rng.integers(0, npod, size=2000))]
        off2 = np.arange(2001, dtype=np.uint32)
        before = e.stats()["keep_route_calls"]
        assert np.array_equal(e.check_bulk_keep_views(e.make_check_views(two), off2).astype(bool), forward(two, off2))
        assert np.array_equal(e.check_bulk_keep_packed(e.make_check_packed(two), off2).astype(bool), forward(two, off2))
        uset = [("pod", names["pod"][int(p_)], "view", "pod", names["pod"][0], "viewer") for p_ in rng.integers(0, npod, size=1000)]
        assert np.array_equal(e.check_bulk_keep_views(e.make_check_views(uset), np.arange(1001, dtype=np.uint32)).astype(bool), forward(uset, np.arange(1001)))
        assert e.stats()["keep_route_calls"] == before
        # an id the API refuses fails the WHOLE call on either route, and the message names the field, the value and the pattern
        bad = list(items[:1200])
        bad[700] = ("pod", "ns3/kube-root-ca.crt", "view", "user", uname, "")
        for call in (lambda: e.check_bulk_keep_views(e.make_check_views(bad), np.arange(1201, dtype=np.uint32)),
                     lambda: e.check_bulk_keep_packed(e.make_check_packed(bad), np.arange(1201, dtype=np.uint32)),
                     lambda: e.check_bulk_views(e.make_check_views(bad)), lambda: e.check_bulk_packed(e.make_check_packed(bad))):
            with pytest.raises(aclgpu.AclError) as ei:
                call()
            msg = str(ei.value)
            assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT and "item 700" in msg and "resource id" in msg and "kube-root-ca.crt" in msg and "byte 16 `.`" in msg, msg
        # the packed request answers as the views do (perm AND err, unknown permission = a pair error, an unknown name = NO_PERMISSION), and refuses a wild index
        mixed = items[:900] + [("pod", names["pod"][5], "nosuchperm", "user", uname, ""), ("pod", "nowhere/x", "view", "user", "ghost", "")]
        pv, ev_ = e.check_bulk_views(e.make_check_views(mixed))
        pp, ep = e.check_bulk_packed(e.make_check_packed(mixed))
        assert np.array_equal(pv, pp) and np.array_equal(ev_, ep) and ep[900] != 0 and pp[901] == 1
        rq, keep_alive = e.make_check_packed(mixed)
        keep_alive[2][3, 1] = rq.n_strings + 5
        with pytest.raises(aclgpu.AclError) as ei:
            e.check_bulk_packed((rq, keep_alive))
        assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT and "dictionary index" in str(ei.value)
        assert routed == 18


def test_postfilter_and_prefilter_mirror(aclgpu):
    """The e2e shape (proxy_test.go:474-531): paul's pods are kept, chani's dropped; items whose template did not
    resolve are kept; prefilter bitmap answers IsAllowed for `ns/name` ids, unknown names are not allowed."""
    from aclgpu import client as v1
    b = kat_runner.load_bootstrap()
    with aclgpu.Engine(b["schema"], "\n".join(b["relationships"])) as e:
        c = v1.PermissionsServiceClient(e)
        mk = lambda pod, u: v1.RelationshipUpdate(v1.OPERATION_TOUCH, v1.Relationship(v1.ObjectReference("pod", pod), "creator",  # noqa: E731
                                                                                       v1.SubjectReference(v1.ObjectReference("user", u))))
        c.WriteRelationships([mk("ns/p1", "paul"), mk("ns/p2", "chani"), mk("ns/p3", "paul"), mk("other/p1", "paul")])
        chk = lambda pod, perm="view": v1.CheckPermissionRequest(v1.ObjectReference("pod", pod), perm,  # noqa: E731
                                                                   v1.SubjectReference(v1.ObjectReference("user", "paul")))
        resolved = [[chk("ns/p1")], [chk("ns/p2")], [chk("ns/p3"), chk("ns/p3", "edit")], None, [chk("ns/p2"), chk("ns/p1")], [], [chk("ns/unknown")]]
        assert v1.filter_items_with_bulk_permissions(c, resolved) == [True, False, True, True, False, True, False]
        pre = v1.PrefilterResult.run_lookup_resources(e, v1.LookupResourcesRequest("pod", "view", v1.SubjectReference(v1.ObjectReference("user", "paul"))))
        assert pre.filter(["ns/p1", "ns/p2", "ns/p3", "other/p1", "nope/nope"]) == [True, False, True, True, False]
        assert pre.is_allowed("ns/p3") and not pre.is_allowed("ns/p2")
        # the same result applied to the kube response's bytes (responsefilterer.go:349-416): list, `kubectl get`'s Table, single object
        import json
        pods = ["ns/p1", "ns/p2", "ns/p3", "other/p1", "nope/nope"]
        md = lambda p: {"namespace": p.split("/")[0], "name": p.split("/")[1]}  # noqa: E731
        body = json.dumps({"kind": "PodList", "items": [{"metadata": md(p), "spec": {}} for p in pods]}).encode()
        out, kept, total = e.prefilter_response("pod", pre.bitmap, "{{namespacedName}}", e.BODY_LIST, body)
        assert [f'{i["metadata"]["namespace"]}/{i["metadata"]["name"]}' for i in json.loads(out)["items"]] == ["ns/p1", "ns/p3", "other/p1"] and (kept, total) == (3, 5)
        table = json.dumps({"kind": "Table", "rows": [{"cells": [p], "object": {"kind": "PartialObjectMetadata", "metadata": md(p)}} for p in pods]}).encode()
        out, kept, _ = e.prefilter_response("pod", pre.bitmap, "{{namespacedName}}", e.BODY_TABLE, table)
        assert [r["cells"][0] for r in json.loads(out)["rows"]] == ["ns/p1", "ns/p3", "other/p1"] and kept == 3
        assert pre.filter_response(table, "table") == out and pre.filter_response(body) == e.prefilter_response("pod", pre.bitmap, "{{namespacedName}}", e.BODY_LIST, body)[0]
        one = json.dumps({"kind": "Pod", "metadata": md("ns/p3")}).encode()
        assert e.prefilter_response("pod", pre.bitmap, "{{namespacedName}}", e.BODY_OBJECT, one)[0] == one
        with pytest.raises(Exception) as x:
            e.prefilter_response("pod", pre.bitmap, "{{namespacedName}}", e.BODY_OBJECT, json.dumps({"metadata": md("ns/p2")}).encode())
        assert x.value.code == 7


def test_micro_batcher_concurrent_single_checks(aclgpu):
    """64 threads x 40 single checks each: every answer equals the oracle's, and the batcher needed fewer device passes
    than there were calls."""
    from aclgpu import workloads
    w = workloads.c1()
    o = orc.Oracle(w.schema)
    w.load(o)
    names = lambda t, n: [f"{t}-{i}" for i in range(n)]  # noqa: E731
    with aclgpu.Engine(w.schema) as e:
        # string ids so that acl_check_one has something to intern: object i is named "<type>-<i>"
        for t, n in w.nobjects.items():
            for nm in names(t, n):
                e.intern(t, nm)
        w.load(e)
        T, PER = 64, 40
        rng = np.random.default_rng(3)
        res = rng.integers(0, w.nobjects["namespace"], size=(T, PER))
        sub = rng.integers(0, w.nobjects["user"], size=(T, PER))
        res[:, ::2] = w.res[rng.integers(0, w.res.size, size=(T, PER // 2))]  # half the stream are known hits' resources
        want, _ = o.check_bulk_ids("namespace", "view", res.reshape(-1), "user", "", sub.reshape(-1))
        got = np.zeros((T, PER), dtype=np.uint8)
        errs = []
        e.batcher_start(max_items=1024, max_wait_us=300)
        passes0 = e.stats()["check_passes"]

        def worker(t):
            try:
                for k in range(PER):
                    p, er = e.check_one("namespace", f"namespace-{res[t, k]}", "view", "user", f"user-{sub[t, k]}")
                    assert er == 0
                    got[t, k] = p
            except BaseException as ex:  # noqa: BLE001
                errs.append(ex)

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        assert not errs, errs[:1]
        st = e.batcher_stats()
        passes = e.stats()["check_passes"] - passes0
        e.batcher_stop()
        assert np.array_equal(got.reshape(-1), want)
        # (Python callers arrive GIL-paced, a few per pass; tools/batcher_bench.cpp measures the coalescing with native threads)
        assert st["items"] == T * PER and passes == st["batches"] and st["batches"] < T * PER, st
        # without the batcher a single check still works (a device pass of its own)
        assert e.check_one("namespace", f"namespace-{res[0, 0]}", "view", "user", f"user-{sub[0, 0]}")[0] == want[0]
        assert e.check_one("", "x", "view", "user", "u") == (0, aclgpu.ERR_INVALID_ARGUMENT)


def test_check_one_submit_and_completions(aclgpu):
    """The non-blocking form of acl_check_one (what a cgo shim binds): 256 logical callers, each with one check outstanding, multiplexed
    over 4 threads that submit and poll.  Every tagged answer equals the oracle's, each completion arrives exactly once, an item that
    fails interning completes with its error, and submitting without a batcher is refused."""
    from aclgpu import workloads
    w = workloads.c1()
    o = orc.Oracle(w.schema)
    w.load(o)
    with aclgpu.Engine(w.schema) as e:
        for t, n in w.nobjects.items():
            for i in range(n):
                e.intern(t, f"{t}-{i}")
        w.load(e)
        with pytest.raises(aclgpu.AclError):
            e.check_one_submit("namespace", "namespace-0", "view", "user", "user-0", tag=1)
        L, PER, M = 256, 12, 4
        rng = np.random.default_rng(11)
        res = rng.integers(0, w.nobjects["namespace"], size=(L, PER))
        sub = rng.integers(0, w.nobjects["user"], size=(L, PER))
        res[:, ::2] = w.res[rng.integers(0, w.res.size, size=(L, PER // 2))]
        want, _ = o.check_bulk_ids("namespace", "view", res.reshape(-1), "user", "", sub.reshape(-1))
        got = np.zeros((L, PER), dtype=np.uint8)
        seen = np.zeros((L, PER), dtype=np.int32)
        progress = np.zeros(L, dtype=np.int64)
        finished = [0]
        lock = threading.Lock()
        errs = []
        e.batcher_start(max_items=1024, max_wait_us=100)

        def submit(t):
            k = int(progress[t])
            progress[t] += 1
            e.check_one_submit("namespace", f"namespace-{res[t, k]}", "view", "user", f"user-{sub[t, k]}", tag=t * 1000 + k)

        def worker(m):
            try:
                for t in range(m, L, M):
                    submit(t)
                while True:
                    with lock:
                        if finished[0] >= L:
                            return
                    for tag, rc, er, p in e.check_completions(64, timeout_s=0.05):
                        t, k = divmod(tag, 1000)
                        assert rc == 0 and er == 0
                        got[t, k] = p
                        seen[t, k] += 1
                        if progress[t] < PER:
                            submit(t)
                        else:
                            with lock:
                                finished[0] += 1
            except BaseException as ex:  # noqa: BLE001
                errs.append(ex)
                with lock:
                    finished[0] = L

        ths = [threading.Thread(target=worker, args=(m,)) for m in range(M)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        assert not errs, errs[:1]
        assert (seen == 1).all()
        assert np.array_equal(got.reshape(-1), want)
        st = e.batcher_stats()
        assert st["items"] == L * PER and st["batches"] < L * PER, st
        # a pair that fails interning completes with its error and never reaches the device
        e.check_one_submit("", "x", "view", "user", "u", tag=77)
        assert e.check_completions(8, timeout_s=2.0) == [(77, 0, aclgpu.ERR_INVALID_ARGUMENT, 0)]
        assert e.check_completions(8, timeout_s=0) == []
        e.batcher_stop()


def test_micro_batcher_coalesces_lookups(aclgpu):
    """Concurrent LookupResources requests (one per list request in the proxy) of the same (type, permission, subject class)
    share batched reverse walks; every set equals the oracle's; requests of another class are walked separately."""
    from tests.test_oracle_cross import SCHEMA
    from tests.test_sharded_gloo import random_tuples
    import random
    tuples = [t for t in random_tuples(random.Random(4), 40) if not (t[0] == t[3] and int(t[4][1:]) <= int(t[1][1:]))]
    co = orc.Oracle(SCHEMA)
    co.write([(orc.OP_TOUCH, t) for t in tuples])
    with aclgpu.Engine(SCHEMA) as e:
        e.write([(aclgpu.OP_TOUCH, t) for t in tuples])
        e.batcher_start(256, 2000)
        reqs = [("doc", "view", "user", f"u{i % 4}", "") for i in range(24)] + [("org", "view", "user", f"u{i % 4}", "") for i in range(8)] + [
            ("doc", "view", "group", "g0", "member")] * 4
        got, errs = [None] * len(reqs), []

        def worker(i):
            try:
                got[i] = e.lookup_one(*reqs[i])
            except BaseException as ex:  # noqa: BLE001
                errs.append(ex)

        ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(reqs))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs[:1]
        for r, g in zip(reqs, got):
            assert g == co.lookup(*r), r
        st = e.batcher_lookup_stats()
        assert st["lookups"] == len(reqs) and 3 <= st["walks"] < len(reqs), st
        # the same requests WITHOUT a blocked thread each (acl_lookup_one_submit / acl_lookup_completions: responsefilterer.go:165-204 starts every
        # prefilter in a goroutine next to the upstream call): tagged answers, engine-allocated rows, one walk per class and sweep
        for i, r in enumerate(reqs):
            e.lookup_one_submit(*r, tag=500 + i)
        e.write([(aclgpu.OP_TOUCH, ("doc", "d-late", "viewer", "user", "u0", ""))])  # objects created after the submits: the rows are sized when the walk runs
        co.write([(orc.OP_TOUCH, ("doc", "d-late", "viewer", "user", "u0", ""))])
        e.lookup_one_submit("doc", "view", "user", "u0", tag=999)
        done = {}
        while len(done) < len(reqs) + 1:
            c = e.lookup_completions(max_items=16, timeout_s=5.0)
            assert c, "lookup completions stopped arriving"
            for tag, rc, cnt, row in c:
                assert rc == 0 and tag not in done
                rt = "doc" if tag == 999 else reqs[tag - 500][0]
                ids = np.flatnonzero(np.unpackbits(row.view(np.uint8), bitorder="little"))
                done[tag] = ({e.object_name(rt, int(k)) for k in ids}, cnt)
        for i, r in enumerate(reqs):
            want = co.lookup(*r)
            assert done[500 + i][0] - {"d-late"} == want - {"d-late"} and done[500 + i][1] == len(done[500 + i][0]), r
        assert done[999][0] == co.lookup("doc", "view", "user", "u0") and "d-late" in done[999][0]
        st2 = e.batcher_lookup_stats()
        assert st2["lookups"] == 2 * len(reqs) + 1 and st2["walks"] - st["walks"] < len(reqs), (st, st2)
        e.batcher_stop()
        assert e.lookup_one("doc", "view", "user", "u1") == co.lookup("doc", "view", "user", "u1")  # no batcher: a walk of its own
        with pytest.raises(aclgpu.AclError):
            e.lookup_one("nosuch", "view", "user", "u1")


def test_watch_then_recheck(aclgpu):
    """RunWatch (watch.go:27-111): per update of the watched type, ONE check of (updated resource, the request's
    subject) decides allowed/denied -- here the polled updates are re-checked as one batch."""
    from aclgpu import client as v1
    b = kat_runner.load_bootstrap()
    with aclgpu.Engine(b["schema"], "\n".join(b["relationships"])) as e:
        c, w = v1.PermissionsServiceClient(e), v1.WatchServiceClient(e)
        recv = w.Watch(["pod"])
        mk = lambda op, pod, rel, u: v1.RelationshipUpdate(op, v1.Relationship(v1.ObjectReference("pod", pod), rel,  # noqa: E731
                                                                                v1.SubjectReference(v1.ObjectReference("user", u))))
        c.WriteRelationships([mk(v1.OPERATION_CREATE, "ns/a", "creator", "paul"), mk(v1.OPERATION_CREATE, "ns/b", "viewer", "chani")])
        c.WriteRelationships([mk(v1.OPERATION_DELETE, "ns/a", "creator", "paul"), mk(v1.OPERATION_TOUCH, "ns/c", "viewer", "paul")])
        events = []
        for resp in recv():
            reqs = [v1.CheckPermissionRequest(v1.ObjectReference("pod", u.relationship.resource.object_id), "view",
                                              v1.SubjectReference(v1.ObjectReference("user", "paul"))) for u in resp.updates]
            pairs = c.CheckBulkPermissions(reqs).pairs
            events += [(u.relationship.resource.object_id, v1.is_allowed(p)) for u, p in zip(resp.updates, pairs)]
        # fully consistent reads: every re-check sees the LATEST state (ns/a's creator is already gone)
        assert events == [("ns/a", False), ("ns/b", False), ("ns/a", False), ("ns/c", True)]
        # ... and the same as ONE entry point (acl_watch_recheck: the poll and the bulk re-check in one call), behind a blocking wait
        _u, cur = e.watch_poll(aclgpu.WATCH_FROM_NOW)
        c.WriteRelationships([mk(v1.OPERATION_TOUCH, "ns/d", "creator", "paul"), mk(v1.OPERATION_TOUCH, "ns/e", "creator", "chani")])
        c.WriteRelationships([mk(v1.OPERATION_TOUCH, "ns/a", "viewer", "paul")])
        assert e.watch_wait(cur, ["pod"], timeout_s=1.0) == e.revision
        got, nxt = e.watch_recheck(cur, "pod", "view", "user", "paul")
        assert [(g[2][1], g[3], g[4]) for g in got] == [("ns/d", 2, 0), ("ns/e", 1, 0), ("ns/a", 2, 0)] and nxt == e.revision
        assert got[0][0] == got[1][0] < got[2][0] and {g[1] for g in got} == {aclgpu.OP_TOUCH}
        assert e.watch_recheck(nxt, "pod", "view", "user", "paul") == ([], nxt)
        with pytest.raises(aclgpu.AclError):
            e.watch_recheck(cur, "nosuchtype", "view", "user", "paul")


def test_concurrent_mixed_calls_are_safe_and_consistent(aclgpu):
    """The seam is called from arbitrary goroutines at once (check.go:77-93, responsefilterer.go:165, workflow workers):
    writers, bulk checkers, single checkers (through the batcher), lookups and watch polls run concurrently; nothing may
    crash or deadlock, every answer must be a legal one, and the final state must equal the oracle's."""
    import random
    from tests.test_oracle_cross import SCHEMA
    rng = random.Random(99)
    users = [f"u{i}" for i in range(6)]
    docs = [f"d{i}" for i in range(12)]
    with aclgpu.Engine(SCHEMA) as e:
        e.write([(aclgpu.OP_TOUCH, ("doc", d, "creator", "user", "u0", "")) for d in docs] +
                [(aclgpu.OP_TOUCH, ("doc", docs[0], "viewer", "group", "g0", "member")), (aclgpu.OP_TOUCH, ("group", "g0", "member", "user", "u1", ""))])
        e.check("doc", docs[0], "view", "user", "u0")  # the snapshot exists before the threads start: later writes must be patched in
        e.batcher_start(256, 100)
        stop = threading.Event()
        errs, applied = [], []
        lock = threading.Lock()
        cursor = [e.watch_poll(aclgpu.WATCH_FROM_NOW)[1]]

        def writer(seed):
            r = random.Random(seed)
            try:
                for _ in range(60):
                    t = ("doc", r.choice(docs), "viewer", "user", r.choice(users[1:]), "")
                    op = r.choice([aclgpu.OP_TOUCH, aclgpu.OP_DELETE])
                    rev = e.write([(op, t)])
                    with lock:
                        applied.append((rev, op, t))
                    time.sleep(0.001)  # let the readers interleave
            except BaseException as ex:  # noqa: BLE001
                errs.append(ex)

        def reader(kind, seed):
            r = random.Random(seed)
            try:
                while not stop.is_set():
                    if kind == "bulk":
                        p, er = e.check_bulk([("doc", r.choice(docs), "view", "user", r.choice(users), "") for _ in range(20)])
                        assert all(x in (1, 2) for x in p) and not any(er)
                    elif kind == "one":
                        p, er = e.check_one("doc", r.choice(docs), "view", "user", r.choice(users))
                        assert p in (1, 2) and er == 0
                        assert e.check_one("doc", r.choice(docs), "view", "user", "u0") == (2, 0)  # creator of every doc, never touched
                    elif kind == "lookup":
                        assert e.lookup("doc", "view", "user", "u0") == set(docs)
                    else:
                        ups, nxt = e.watch_poll(cursor[0], ["doc"])
                        assert all(u[2][0] == "doc" for u in ups) and nxt >= cursor[0]
                        cursor[0] = nxt
            except BaseException as ex:  # noqa: BLE001
                errs.append(ex)

        ws = [threading.Thread(target=writer, args=(s,)) for s in range(3)]
        rs = [threading.Thread(target=reader, args=(k, i)) for i, k in enumerate(["bulk", "one", "one", "lookup", "watch", "bulk"])]
        for t in ws + rs:
            t.start()
        for t in ws:
            t.join(timeout=120)
        stop.set()
        for t in rs:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ws + rs), "deadlock"
        assert not errs, errs[:2]
        e.batcher_stop()
        # replay the committed history (revision order) into the oracle: final answers must agree
        co = orc.Oracle(SCHEMA)
        co.write([(orc.OP_TOUCH, ("doc", d, "creator", "user", "u0", "")) for d in docs] +
                 [(orc.OP_TOUCH, ("doc", docs[0], "viewer", "group", "g0", "member")), (orc.OP_TOUCH, ("group", "g0", "member", "user", "u1", ""))])
        for _rev, op, t in sorted(applied):
            co.write([(orc.OP_TOUCH if op == aclgpu.OP_TOUCH else orc.OP_DELETE, t)])
        qs = [("doc", d, "view", "user", u, "") for d in docs for u in users]
        perms, errs2 = e.check_bulk(qs)
        assert list(zip(perms, errs2)) == [co.check(*q) for q in qs]
        for u in users:
            assert e.lookup("doc", "view", "user", u) == co.lookup("doc", "view", "user", u)
        assert len(applied) == 180 and e.stats()["snapshot_patches"] > 0


def test_cancellation_and_deadline(aclgpu):
    """LookupResources runs on the HTTP request's context and is abandoned when that is cancelled
    (responsefilterer.go:165-170); the prefilter join gives up after 10 s (responsefilterer.go:44,196-204).  The C side of a
    context.Context is acl_call_opts_t: a cancel flag the engine polls and a timeout."""
    import ctypes as C
    import threading
    import time
    from aclgpu import workloads
    w = workloads.c4(scale=0.02, batch=20000, n_user=20000)
    with aclgpu.Engine(w.schema) as e:
        w.load(e)
        items = e.make_items("pod", "view", w.res, "user", "", w.subj)
        want = e.check_bulk_ids(items)
        # already cancelled / already past its deadline: refused before any device work, with the gRPC codes the shim forwards
        flag = C.c_int32(1)
        with pytest.raises(aclgpu.AclError) as ei:
            e.check_bulk_ids_opts(items, cancel=flag)
        assert ei.value.code == aclgpu.ERR_CANCELLED
        with pytest.raises(aclgpu.AclError) as ei:
            e.check_bulk_ids_opts(items, timeout_s=1e-9)
        assert ei.value.code == aclgpu.ERR_DEADLINE_EXCEEDED
        e.intern("user", "somebody")
        with pytest.raises(aclgpu.AclError) as ei:
            e.lookup_one("pod", "view", "user", "somebody", cancel=flag)
        assert ei.value.code == aclgpu.ERR_CANCELLED
        # not cancelled, generous deadline: same answers
        flag.value = 0
        p, er = e.check_bulk_ids_opts(items, cancel=flag, timeout_s=30.0)
        assert np.array_equal(p, want[0]) and np.array_equal(er, want[1])
        # cancelled while parked behind the micro-batcher's window: the caller returns, the batch still completes for the others
        e.batcher_start(max_items=1 << 20, max_wait_us=200_000)
        try:
            out = {}

            def blocked():
                try:
                    out["r"] = e.check_one("pod", "nope", "view", "user", "nobody", cancel=flag)
                except aclgpu.AclError as ex:
                    out["code"] = ex.code
                out["t"] = time.perf_counter()

            t0 = time.perf_counter()
            th = threading.Thread(target=blocked)
            th.start()
            time.sleep(0.02)
            flag.value = 1  # ctx.Done()
            th.join(5)
            assert out.get("code") == aclgpu.ERR_CANCELLED and out["t"] - t0 < 0.15, out  # well before the 200 ms window closes
            flag.value = 0
            assert e.check_one("pod", "nope", "view", "user", "nobody", timeout_s=5.0) == (1, 0)
            time.sleep(0.3)  # (every dispatcher idle again: a window left over from the calls above may legitimately end early -- max_wait is an upper bound)
            with pytest.raises(aclgpu.AclError) as ei:
                e.check_one("pod", "nope", "view", "user", "nobody", timeout_s=0.01)  # deadline inside the 200 ms window
            assert ei.value.code == aclgpu.ERR_DEADLINE_EXCEEDED
        finally:
            e.batcher_stop()


def test_every_call_shape_at_once_native_threads(aclgpu, tmp_path):
    """tools/engine_stress.cpp: three blocking callers with chip-filling batches (chained on the device), a submit/wait window, 64-item
    batches, single checks (blocking and completion queue), string batches through the interning pool (one caller back to back, one that finds the
    pool asleep), PostFilter calls for one user (the reverse-walk route), LookupResources and a writer that forces snapshot patches -- all at
    once on one engine, every answer compared with the same call made alone.  (Found the pipeline's look-ahead staging deadlocking against a queued
    writer: Eval::begin with try_only must not wait for the state lock.)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "spicedb-kubeapi-proxy_amd", "lib")
    exe = tmp_path / "engine_stress"
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(root, "tools", "engine_stress.cpp"), "-I", os.path.join(root, "include"), "-L", lib,
                           "-laclgpu", "-lpthread", f"-Wl,-rpath,{lib}", "-o", str(exe)])
    pr = subprocess.run([str(exe), "2"], capture_output=True, text=True, timeout=120)
    assert pr.returncode == 0, (pr.stdout + pr.stderr)[-600:]
    assert " 0 wrong or failed" in pr.stdout, pr.stdout



def test_check_bulk_of_one_subject_behind_a_cycle_of_groups_carries_the_depth_errors(aclgpu):
    """CheckBulkPermissions by one reverse walk (engine.cpp keep_by_reverse_walk, pair form) where the row's missing bit cannot tell NO_PERMISSION from "gave up at
    the dispatch-depth limit": `group#member` is recursive and two groups contain each other, so pods shared with them answer a depth ERROR for a user who is in
    neither -- per pair, as the oracle does.  The depth sweep (no_object_is_deep) finds those pods and hands their bitmap to the route: the first call goes forward
    (nothing known about the snapshot yet), the second sweeps and walks, the third walks -- every pair's permissionship AND error the oracle's each time; the keep
    call, which drops an item on either answer, walks from the start."""
    schema = """definition user {}
definition group { relation member: user | group#member }
definition pod { relation viewer: user | group#member
 permission view = viewer }"""
    rels = ["group:g1#member@group:g2#member", "group:g2#member@group:g1#member", "group:g3#member@user:u1"]
    rels += [f"pod:p{i}#viewer@group:g1#member" for i in range(0, 700, 3)] + [f"pod:p{i}#viewer@user:u1" for i in range(1, 700, 3)] + [f"pod:p{i}#viewer@group:g3#member" for i in range(2, 700, 3)]
    o = orc.Oracle(schema)
    o.write([(orc.OP_TOUCH, r) for r in rels])
    items = [("pod", f"p{i}", "view", "user", "u1", "") for i in range(700)]
    want = [o.check(*q) for q in items]
    assert {w_[1] for w_ in want} == {0, aclgpu.ERR_DEPTH} and sum(w_[0] == 2 for w_ in want) > 400
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        before = e.stats()["keep_route_calls"]
        for got in (e.check_bulk_views(e.make_check_views(items)), e.check_bulk_packed(e.make_check_packed(items)), e.check_bulk_views(e.make_check_views(items))):
            assert list(zip(got[0].tolist(), got[1].tolist())) == want
        # (the second call at the same snapshot swept the type and found the pods behind the cycle: it and the third answered by the walk + their bitmap)
        assert e.stats()["keep_route_calls"] == before + 2 and e.stats()["depth_sweeps"] == 1
        for u in ("nobody", "u1"):  # a subject no table knows: no walk at all, NO_PERMISSION everywhere but behind the cycle
            qs = [q[:4] + (u, "") for q in items]
            got = e.check_bulk_views(e.make_check_views(qs))
            assert list(zip(got[0].tolist(), got[1].tolist())) == [o.check(*q) for q in qs], u
        before = e.stats()["keep_route_calls"]
        keep = e.check_bulk_keep_views(e.make_check_views(items), np.arange(701, dtype=np.uint32)).astype(bool)
        assert e.stats()["keep_route_calls"] == before + 1 and keep.tolist() == [w_ == (2, 0) for w_ in want]



def test_check_bulk_of_one_subject_on_nested_groups_takes_the_reverse_walk_once_the_snapshot_is_known_shallow(aclgpu):
    """The pair form on a RECURSIVE permission (nested groups, SURVEY 8(d) C4's schema): the schema cannot rule a depth error out, the snapshot can -- one forward
    sweep over the type for a subject nobody is (engine.cpp no_object_is_deep; the property it rests on: tests/test_oracle_cross.py
    test_a_depth_error_does_not_depend_on_the_subject).  First call at a snapshot: forward, and a note; second: the sweep, then the reverse walk; a write starts
    over.  A cycle written behind some pods: after one more sweep the calls still walk, their pairs' depth errors from the deep pods' bitmap (as the oracle's); deleting it: shallow again.
    Every answer is the oracle's (check.go:54-69: pair i answers item i)."""
    schema = """definition user {}
definition group { relation member: user | group#member }
definition namespace { relation viewer: user | group#member
 permission view = viewer }
definition pod { relation namespace: namespace
 relation viewer: user | group#member
 permission view = viewer + namespace->view }"""
    rels = [f"group:l{d}-{k}#member@group:l{d + 1}-{(2 * k + j) % 8}#member" for d in range(4) for k in range(8) for j in range(2)]
    rels += [f"group:l4-{k}#member@user:u{k}" for k in range(8)] + ["group:l2-3#member@user:mid"]
    rels += [f"namespace:n{k}#viewer@group:l1-{k}#member" for k in range(8)]
    rels += [f"pod:p{i}#namespace@namespace:n{i % 16}" for i in range(900)]  # (n8..n15 have no viewers)
    rels += [f"pod:p{i}#viewer@group:l0-{i % 8}#member" for i in range(0, 900, 5)] + [f"pod:p{i}#viewer@user:u{i % 8}" for i in range(1, 900, 7)]
    o = orc.Oracle(schema)
    for b in range(0, len(rels), 1000):  # (spicedb.go:35: at most 1 000 updates per write)
        o.write([(orc.OP_TOUCH, r) for r in rels[b:b + 1000]])
    users = ["u0", "u5", "mid", "stranger"]
    items = {u: [("pod", f"p{i}", "view", "user", u, "") for i in range(900)] + [("pod", "no-such-pod", "view", "user", u, "")] for u in users}

    def answers(e, u, packed=False):
        got = e.check_bulk_packed(e.make_check_packed(items[u])) if packed else e.check_bulk_views(e.make_check_views(items[u]))
        return list(zip(got[0].tolist(), got[1].tolist()))

    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        want = {u: [o.check(*q) for q in items[u]] for u in users}
        assert all(w_[1] == 0 for u in users for w_ in want[u]) and 100 < sum(w_[0] == 2 for w_ in want["u0"]) < 900
        st0 = e.stats()
        assert answers(e, "u0") == want["u0"]  # forward: nothing is known about this snapshot yet
        st1 = e.stats()
        assert (st1["keep_route_calls"], st1["depth_sweeps"]) == (st0["keep_route_calls"], st0["depth_sweeps"])
        for k, u in enumerate(users):  # the sweep (once), then the walk -- for every user, the unknown one too
            assert answers(e, u, packed=bool(k & 1)) == want[u], u
        st2 = e.stats()
        assert st2["depth_sweeps"] == st1["depth_sweeps"] + 1 and st2["keep_route_calls"] == st1["keep_route_calls"] + len(users)
        # ---- plain grants and removals end paths or take them away (Store::path_adds): the snapshot changes, what the sweep showed still holds -- the very
        # next call walks, and answers for the store as it is now
        for op, rel in ((aclgpu.OP_TOUCH, "pod:p3#viewer@user:stranger"), (aclgpu.OP_DELETE, "pod:p1#viewer@user:u1"), (aclgpu.OP_DELETE, "group:l1-0#member@group:l2-0#member")):
            e.write([(op, rel)])
            o.write([(op, rel)])
            want = {u: [o.check(*q) for q in items[u]] for u in users}
            for u in users:
                assert answers(e, u) == want[u], (rel, u)
        assert want["stranger"][3] == (2, 0)
        st2b = e.stats()
        assert st2b["depth_sweeps"] == st2["depth_sweeps"] and st2b["keep_route_calls"] == st2["keep_route_calls"] + 3 * len(users)
        # ... a new nesting edge may add a path: forward once, then a sweep
        for op, rel in ((aclgpu.OP_TOUCH, "group:l1-0#member@group:l2-0#member"),):
            e.write([(op, rel)])
            o.write([(op, rel)])
        want = {u: [o.check(*q) for q in items[u]] for u in users}
        for u in users:
            assert answers(e, u) == want[u], u
        st2 = e.stats()
        assert st2["depth_sweeps"] == st2b["depth_sweeps"] + 1 and st2["keep_route_calls"] == st2b["keep_route_calls"] + len(users) - 1
        # ---- a cycle behind the l3 groups: pods that reach it answer a depth error for whoever is not found first
        cyc = ["group:l4-1#member@group:l3-0#member"]
        e.write([(aclgpu.OP_TOUCH, cyc[0])])
        o.write([(orc.OP_TOUCH, cyc[0])])
        want = {u: [o.check(*q) for q in items[u]] for u in users}
        assert any(w_[1] == aclgpu.ERR_DEPTH for w_ in want["stranger"]) and any(w_ == (2, 0) for w_ in want["u0"])
        for rnd in range(3):
            for u in users:
                assert answers(e, u) == want[u], (rnd, u)
        st3 = e.stats()
        # (a new path: the first call forward, the second sweeps and finds deep pods; it and the ten after it walk, the depth errors from the pods' bitmap)
        assert st3["depth_sweeps"] == st2["depth_sweeps"] + 1 and st3["keep_route_calls"] == st2["keep_route_calls"] + 3 * len(users) - 1
        # ---- the cycle deleted: a new snapshot, shallow again
        e.write([(aclgpu.OP_DELETE, cyc[0])])
        o.write([(orc.OP_DELETE, cyc[0])])
        want = {u: [o.check(*q) for q in items[u]] for u in users}
        for rnd in range(2):
            for u in users:
                assert answers(e, u) == want[u], (rnd, u)
        st4 = e.stats()
        assert st4["depth_sweeps"] == st3["depth_sweeps"] + 1 and st4["keep_route_calls"] == st3["keep_route_calls"] + 2 * len(users) - 1


def test_postfilter_k_items_times_f_templates_one_walk_per_template(aclgpu):
    """K list items x F PostFilter templates for the requesting user (postfilter.go:86-119: every filter of every matching rule): pair j of every item comes
    from template j, so each template position is a one-subject call of its own -- engine.cpp keep_by_reverse_walks answers the F of them by their reverse walks
    under ONE evaluation and ANDs the masks.  Compared with the oracle (an item is kept when every pair is HAS_PERMISSION without error, postfilter.go:144-178);
    the stats say F walks were taken.  Shapes the route must leave to the forward path: an item with a pair missing, a template position with two subjects."""
    schema = """definition user {}
definition group { relation member: user | group#member }
definition namespace { relation viewer: user | group#member
 permission view = viewer }
definition pod { relation namespace: namespace
 relation viewer: user | group#member
 relation creator: user
 permission view = viewer + creator + namespace->view
 permission edit = creator }"""
    rels = [f"group:g{k}#member@user:u{k}" for k in range(6)] + ["group:g0#member@group:g1#member"]
    rels += [f"namespace:n{k}#viewer@group:g{k % 6}#member" for k in range(12)]
    rels += [f"pod:n{i % 16}/p{i}#namespace@namespace:n{i % 16}" for i in range(800)]
    rels += [f"pod:n{i % 16}/p{i}#creator@user:u{i % 7}" for i in range(0, 800, 2)] + [f"pod:n{i % 16}/p{i}#viewer@user:u{(i * 5) % 7}" for i in range(1, 800, 3)]
    o = orc.Oracle(schema)
    for b in range(0, len(rels), 1000):
        o.write([(orc.OP_TOUCH, r) for r in rels[b:b + 1000]])
    K = 800
    with aclgpu.Engine(schema, "\n".join(rels)) as e:
        for user in ("u1", "u0", "nobody"):
            for tpls in ((("pod", "{pod}", "view"), ("pod", "{pod}", "edit")), (("pod", "{pod}", "view"), ("namespace", "{ns}", "view"), ("pod", "{pod}", "edit"))):
                F = len(tpls)
                pairs = [(t, rid.format(pod=f"n{i % 16}/p{i}", ns=f"n{i % 16}"), perm, "user", user, "") for i in range(K) for t, rid, perm in tpls]
                off = np.arange(0, F * K + 1, F, dtype=np.uint32)
                want = [all(o.check(*pairs[i * F + j]) == (2, 0) for j in range(F)) for i in range(K)]
                before = e.stats()["keep_route_calls"]
                for keep in (e.check_bulk_keep_views(e.make_check_views(pairs), off), e.check_bulk_keep_packed(e.make_check_packed(pairs), off)):
                    assert keep.astype(bool).tolist() == want, (user, F)
                assert e.stats()["keep_route_calls"] == before + 2 * F, (user, F)
                assert user == "nobody" or 0 < sum(want) < K
        # ---- not this route: item 5 lacks its second pair; position 1 names two subjects -- same masks by the forward path
        tpls = (("pod", "{pod}", "view"), ("pod", "{pod}", "edit"))
        pairs = [(t, rid.format(pod=f"n{i % 16}/p{i}"), perm, "user", "u1", "") for i in range(K) for t, rid, perm in tpls]
        ragged = pairs[:11] + pairs[12:]
        off = np.array([0] + [2 * i + 2 - (i >= 5) for i in range(K)], dtype=np.uint32)
        before = e.stats()["keep_route_calls"]
        keep = e.check_bulk_keep_views(e.make_check_views(ragged), off).astype(bool).tolist()
        assert keep == [all(o.check(*ragged[j]) == (2, 0) for j in range(off[i], off[i + 1])) for i in range(K)]
        two = list(pairs)
        two[2 * 400 + 1] = two[2 * 400 + 1][:4] + ("u2", "")
        keep = e.check_bulk_keep_views(e.make_check_views(two), np.arange(0, 2 * K + 1, 2, dtype=np.uint32)).astype(bool).tolist()
        assert keep == [all(o.check(*two[2 * i + j]) == (2, 0) for j in range(2)) for i in range(K)]
        assert e.stats()["keep_route_calls"] == before
