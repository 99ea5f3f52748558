"""CPU-only host logic: the engine's relationship store (the write side of the seam:
WriteRelationships / ReadRelationships / DeleteRelationships, preconditions, expiration)
replayed against the golden vectors, with the oracle as the checker for store contents.
Check / lookup steps need the GPU and are exercised by tests/test_engine_gpu.py."""
import pytest

from oracle import orc
from tests import kat_runner

KATS = kat_runner.load_kats()


class StoreOnly:
    """kat_runner adapter that replays only the store-side steps on the engine and mirrors
    them into the oracle, comparing the full relationship listing after every step."""

    def __init__(self, engine, oracle):
        self.e, self.o = engine, oracle

    def write(self, ups, pre):
        err_o = err_e = None
        try:
            self.o.write(ups, pre)
        except orc.OracleError as x:
            err_o = x.code
        try:
            self.e.write(ups, pre)
        except Exception as x:  # noqa: BLE001
            err_e = x.code
        assert err_o == err_e, (ups, pre, err_o, err_e)
        if err_e:
            e = Exception("write failed")
            e.code = err_e
            raise e

    def read(self, **f):
        a, b = sorted(self.e.read(**f)), sorted(self.o.read(**f))
        assert a == b
        return a

    def delete_by_filter(self, **f):
        n1, n2 = self.e.delete_by_filter(**f), self.o.delete_by_filter(**f)
        assert n1 == n2
        return n1

    def set_now(self, t):
        self.e.set_now(t)
        self.o.set_now(t)

    def check(self, *a):
        return self.o.check(*a)  # GPU-only in the engine; keep the KAT flowing with the checker's answer

    def lookup(self, *a):
        return self.o.lookup(*a)


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_store_kat(kat, aclgpu_lib):
    import aclgpu
    schema, rels = kat_runner.kat_schema(kat)
    e = aclgpu.Engine(schema, "\n".join(rels), store_only=True)
    o = orc.Oracle(schema)
    if rels:
        o.write([(orc.OP_TOUCH, r) for r in rels])
    types = [t for t in ("namespace", "pod", "testresource", "lock", "workflow", "group", "cluster", "user") if e.type_id(t) >= 0]
    ad = StoreOnly(e, o)
    kat_runner.run_kat(kat, ad)
    for t in types:
        assert sorted(e.read(rtype=t)) == sorted(o.read(rtype=t)), t


@pytest.mark.parametrize("good", [
    "definition a { relation r: a | b:* }\ndefinition b {}",
    "definition u {}\ndefinition a { relation r: u\n permission p = r & r }",
    "definition u {}\ndefinition a { relation r: u\n permission p = r - r }",
    "definition u {}\ndefinition a { relation r: u | u:*\n relation q: a\n permission p = (r - q->p) & r + nil }",
    "definition u {}\ndefinition a { relation r: a\n relation v: u\n permission p = v + r.all(p) - r.any(p) }",  # the intersection arrow
])
def test_engine_accepts_round4_schema_features(good, aclgpu_lib):
    """intersection, exclusion, wildcard subjects: the reference boots any schema (pkg/spicedb/spicedb.go:19-24)"""
    import aclgpu
    aclgpu.Engine(good, store_only=True).close()


def test_wildcard_relationships_in_the_store(aclgpu_lib):
    """`T:*` is a subject class of its own: written, read back with subject id "*", refused where the relation does not allow it."""
    import aclgpu
    e = aclgpu.Engine("definition user {}\ndefinition doc { relation viewer: user | user:*\n relation editor: user }", store_only=True)
    e.write([(aclgpu.OP_TOUCH, ("doc", "d1", "viewer", "user", "*", "")), (aclgpu.OP_TOUCH, ("doc", "d1", "viewer", "user", "alice", ""))])
    assert sorted(r[4] for r in e.read(rtype="doc", rid="d1")) == ["*", "alice"]
    assert [r[4] for r in e.read(rtype="doc", stype="user", sid="*")] == ["*"]
    for bad in [("doc", "d1", "editor", "user", "*", ""), ("doc", "d1", "viewer", "user", "*", "viewer")]:
        with pytest.raises(aclgpu.AclError):
            e.write([(aclgpu.OP_TOUCH, bad)])
    with pytest.raises(aclgpu.AclError):  # CREATE of an existing wildcard relationship conflicts like any other
        e.write([(aclgpu.OP_CREATE, ("doc", "d1", "viewer", "user", "*", ""))])
    e.write([(aclgpu.OP_DELETE, ("doc", "d1", "viewer", "user", "*", ""))])
    assert [r[4] for r in e.read(rtype="doc", rid="d1")] == ["alice"]
    e.close()


@pytest.mark.parametrize("bad", [
    "definition a { relation r: a | b:x }\ndefinition b {}",
    "definition u {}\ndefinition a { relation r: u | u:*\n relation q: a\n permission p = r->q }",
    "definition u {}\ndefinition a { relation r: u\n permission p = r & }",
    "caveat c(x int) { x > 1 }\ndefinition u {}",
    "definition u {}\ndefinition a { relation r: u with c }",
    "definition u {}\ndefinition a { relation r: a\n permission p = r.some(p) }",  # (only .any() and .all() are arrow functions)
    "definition u {}\ndefinition b {}\ndefinition a { relation r: a | b\n relation v: u\n permission p = v + r.all(p) }",  # .all() over a subject type without the permission: refused (fails closed)
    "definition a { relation r: nosuch }",
    "definition u {}\ndefinition a { relation r: u\n permission p = nosuch }",
    "definition u {}\ndefinition u {}",
])
def test_engine_rejects_unsupported_schema(bad, aclgpu_lib):
    import aclgpu
    with pytest.raises(aclgpu.AclError) as ei:
        aclgpu.Engine(bad, store_only=True)
    assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT


def test_write_limits(aclgpu_lib):
    """<= 1000 updates per write: reference pkg/spicedb/spicedb.go:35-36."""
    import aclgpu
    e = aclgpu.Engine("definition user {}\ndefinition doc { relation viewer: user }", store_only=True)
    ups = [(aclgpu.OP_TOUCH, ("doc", f"d{i}", "viewer", "user", "u", "")) for i in range(1001)]
    with pytest.raises(aclgpu.AclError) as ei:
        e.write(ups)
    assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT
    e.write(ups[:1000])
    assert len(e.read(rtype="doc")) == 1000


def test_bulk_edges_and_text_agree(aclgpu_lib):
    import aclgpu
    import numpy as np
    e = aclgpu.Engine("definition user {}\ndefinition doc { relation viewer: user }", store_only=True)
    e.add_edges("doc", "viewer", "user", "", np.array([3, 1, 3, 1], dtype=np.uint32), np.array([7, 2, 7, 9], dtype=np.uint32))
    got = sorted((r[1], r[4]) for r in e.read(rtype="doc"))
    assert got == [("#1", "#2"), ("#1", "#9"), ("#3", "#7")]  # anonymous ids print as #n; duplicates collapse (TOUCH)
    assert e.object_count("doc") == 4 and e.object_count("user") == 10


def test_watch_feed(aclgpu_lib):
    """Change feed behind WatchService.Watch (pkg/authz/watch.go:29-38): commit order, type filter, cursor, and the
    client mirror that groups updates per revision."""
    import aclgpu
    from aclgpu import client as v1
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
    ups, cur = e.watch_poll(aclgpu.WATCH_FROM_NOW)
    assert ups == [] and cur == e.revision  # bootstrap relationships are not part of the feed
    w = v1.WatchServiceClient(e)
    recv_pods, recv_all = w.Watch(["pod"]), w.Watch()
    c = v1.PermissionsServiceClient(e)
    rel = lambda t, i, r, u: v1.Relationship(v1.ObjectReference(t, i), r, v1.SubjectReference(v1.ObjectReference("user", u)))  # noqa: E731
    c.WriteRelationships([v1.RelationshipUpdate(v1.OPERATION_CREATE, rel("pod", "ns/p1", "creator", "paul")),
                          v1.RelationshipUpdate(v1.OPERATION_TOUCH, rel("namespace", "ns", "creator", "paul"))])
    c.WriteRelationships([v1.RelationshipUpdate(v1.OPERATION_DELETE, rel("pod", "ns/p1", "creator", "paul")),
                          v1.RelationshipUpdate(v1.OPERATION_CREATE, rel("pod", "ns/p1", "creator", "chani"))])
    got = recv_pods()
    assert [len(r.updates) for r in got] == [1, 2] and got[0].changes_through < got[1].changes_through
    assert [(u.operation, u.relationship.resource.object_id, u.relationship.subject.object.object_id) for r in got for u in r.updates] == [
        (v1.OPERATION_TOUCH, "ns/p1", "paul"), (v1.OPERATION_DELETE, "ns/p1", "paul"), (v1.OPERATION_TOUCH, "ns/p1", "chani")]
    assert recv_pods() == []  # nothing new
    assert sum(len(r.updates) for r in recv_all()) == 4
    assert c.DeleteRelationships(v1.RelationshipFilter("pod", "ns/p1")).relationships_deleted_count == 1
    (r,) = recv_pods()
    assert [(u.operation, u.relationship.subject.object.object_id) for u in r.updates] == [(v1.OPERATION_DELETE, "chani")]
    with pytest.raises(aclgpu.AclError):
        e.watch_poll(0, ["nosuchtype"])
    e.load_bootstrap(b["schema"])  # a new schema invalidates old cursors
    with pytest.raises(aclgpu.AclError) as ei:
        e.watch_poll(cur)
    assert ei.value.code == aclgpu.ERR_OUT_OF_RANGE
    e.close()


def test_watch_wait_blocks_until_the_feed_moves(aclgpu_lib):
    """acl_watch_wait: the blocking half of Watch.Recv (pkg/authz/watch.go:38) -- a condition variable on the write path instead of a sleeping
    poll per stream.  It wakes for updates of the watched types only, honours the deadline and the cancel flag, and an update committed BEFORE
    the wait started is seen without waiting."""
    import ctypes
    import threading
    import time
    import aclgpu
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
    _ups, cur = e.watch_poll(aclgpu.WATCH_FROM_NOW)
    t0 = time.perf_counter()
    with pytest.raises(aclgpu.AclError) as ei:  # nothing happens: the deadline
        e.watch_wait(cur, ["pod"], timeout_s=0.05)
    assert ei.value.code == aclgpu.ERR_DEADLINE_EXCEEDED and 0.04 < time.perf_counter() - t0 < 1.0
    # a write to ANOTHER type does not end a wait for pods; a pod write does, promptly
    woke = []

    def waiter():
        t1 = time.perf_counter()
        woke.append((e.watch_wait(cur, ["pod"], timeout_s=5.0), time.perf_counter() - t1))

    th = threading.Thread(target=waiter)
    th.start()
    time.sleep(0.05)
    e.write([(aclgpu.OP_TOUCH, ("namespace", "other", "viewer", "user", "paul", ""))])
    time.sleep(0.05)
    assert not woke
    e.write([(aclgpu.OP_TOUCH, ("pod", "ns/p1", "creator", "user", "paul", ""))])
    th.join(timeout=5)
    assert woke and woke[0][0] == e.revision and woke[0][1] < 1.0
    ups, nxt = e.watch_poll(cur, ["pod"])
    assert [u[2][1] for u in ups] == ["ns/p1"]
    assert e.watch_wait(cur, ["pod"], timeout_s=0.01) == e.revision  # already there: no wait
    # delete-by-filter wakes waiters too; the cancel flag ends a wait that has no deadline
    th2_out = []
    th2 = threading.Thread(target=lambda: th2_out.append(e.watch_wait(nxt, ["pod"], timeout_s=5.0)))
    th2.start()
    time.sleep(0.03)
    assert e.delete_by_filter(rtype="pod", rid="ns/p1") == 1
    th2.join(timeout=5)
    assert th2_out == [e.revision]
    flag = ctypes.c_int32(0)
    res = []

    def cancelled():
        try:
            e.watch_wait(e.revision, [], cancel=flag)
        except aclgpu.AclError as x:
            res.append(x.code)

    th3 = threading.Thread(target=cancelled)
    th3.start()
    time.sleep(0.03)
    flag.value = 1
    th3.join(timeout=5)
    assert res == [aclgpu.ERR_CANCELLED]
    with pytest.raises(aclgpu.AclError):
        e.watch_wait(cur, ["nosuchtype"], timeout_s=0.01)
    # the bulk re-check of a poll needs the device; a poll without updates does not
    assert e.watch_recheck(e.revision, "pod", "view", "user", "paul") == ([], e.revision)
    e.write([(aclgpu.OP_TOUCH, ("pod", "ns/p2", "creator", "user", "paul", ""))])
    with pytest.raises(aclgpu.AclError) as ei:
        e.watch_recheck(nxt, "pod", "view", "user", "paul")
    assert ei.value.code == aclgpu.ERR_UNAVAILABLE
    e.close()


def test_keep_and_check_one_need_a_gpu(aclgpu_lib):
    import aclgpu
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], store_only=True)
    with pytest.raises(aclgpu.AclError) as ei:
        e.check_bulk_keep([("namespace", "a", "view", "user", "u", "")], [0, 1])
    assert ei.value.code == aclgpu.ERR_UNAVAILABLE
    with pytest.raises(aclgpu.AclError):
        e.check_one("namespace", "a", "view", "user", "u")
    # an empty request never reaches the device: the pair carries InvalidArgument (options_test.go:101-102)
    assert e.check_one("", "", "", "", "") == (0, aclgpu.ERR_INVALID_ARGUMENT)
    e.close()


def test_object_names_of_every_length_round_trip(aclgpu_lib):
    """The name table keeps a name's first 46 bytes inside its hash slot and the rest outside (store.hpp ObjectTable::Slot): names around
    that length, names that share a 46-byte prefix, prefixes of stored names and one far beyond 65 535 bytes (the slot's length field
    saturates) must all intern to distinct ids, be found again, and come back byte for byte -- across the table's growth steps."""
    import aclgpu
    e = aclgpu.Engine("definition user {}", store_only=True)
    base = "ns-0123456789/pod-abcdefghijklmnopqrstuvwxyz-0123456789-ABCDEFGHIJKLMNOPQRSTUVWXYZ"
    names = [base[:k] for k in (1, 2, 15, 16, 45, 46, 47, 48, 63, 64, 65, len(base))]
    names += [base[:46] + s for s in ("x", "y", "xx", "x" * 30)]          # same inline prefix, different tails
    names += ["y" * 23, "y" * 24, "z" * 1022, "z" * 1023, "z" * 1024]     # up to the API's 1024-byte limit (validate.hpp)
    names += [f"filler-{i}" for i in range(3000)]                          # several growth steps: slots are re-hashed with their names
    ids = [e.intern("user", n) for n in names]
    assert ids == list(range(len(names)))
    assert [e.intern("user", n) for n in names] == ids  # idempotent
    assert [e.find("user", n) for n in names] == ids
    for n in (base[:44], base[:46] + "z", base + "!", "z" * 1021, "z" * 1025, "filler-3000", ""):
        assert e.find("user", n) is None, n[:60]
    for bad in ("é" * 23, "z" * 1025, "a.b", "a b", "*", "a:b"):  # what the API's object-id pattern refuses never enters a table
        with pytest.raises(aclgpu.AclError) as ei:
            e.intern("user", bad)
        assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT
    assert [e.object_name("user", i) for i in ids[:21]] == names[:21]


def test_object_id_pattern_byte_by_byte(aclgpu_lib):
    """validate.hpp checks an object id 16 bytes per step (two overlapping 8- or 4-byte loads below 16 bytes, an overlapping last block above): every byte
    value 1..255 at the first, middle and last position of ids of every length 1..40 -- each block shape, each lane -- must be judged as the API's pattern
    `^[a-zA-Z0-9/_|\\-=+]{1,}$` judges it (UNVERIFIED restatement, validate.hpp header).  NUL cannot be passed through a C string and ends the id."""
    import ctypes
    import re
    import aclgpu
    e = aclgpu.Engine("definition user {}", store_only=True)
    pat = re.compile(rb"[a-zA-Z0-9/_|\-=+]+")
    tid, out = e.type_id("user"), ctypes.c_uint32()
    wrong = []
    for n in range(1, 41):
        for pos in sorted({0, n // 2, n - 1, min(n - 1, 7), min(n - 1, 8), min(n - 1, 15), min(n - 1, 16)}):
            for c in range(1, 256):
                b = bytearray(b"q" * n)
                b[pos] = c
                rc = e._L.acl_intern(e._h, tid, bytes(b), ctypes.byref(out))
                want = pat.fullmatch(bytes(b)) is not None
                if (rc == 0) != want or (rc != 0 and rc != aclgpu.ERR_INVALID_ARGUMENT):
                    wrong.append((n, pos, c, rc))
    assert not wrong, wrong[:10]
    e.close()


def test_bootstrap_yaml_files(aclgpu_lib):
    """acl_load_bootstrap_yaml: the YAML form the reference boots from (the embedded pkg/spicedb/bootstrap.yaml:1-40, a path in the endpoint
    URL options.go:313-316, a byte map spicedb.go:19-21).  The reference's own default file, re-indented and re-chomped, comments, several
    documents, keys the proxy does not use; what is outside the supported subset fails loudly."""
    import aclgpu
    b = kat_runner.load_bootstrap()
    indent = lambda text, n: "\n".join((" " * n + ln) if ln.strip() else "" for ln in text.split("\n"))  # noqa: E731
    default = "schema: |-\n" + indent(b["schema"], 2) + "\nrelationships: |-\n" + indent("\n".join(b["relationships"]), 2) + "\n"
    e = aclgpu.Engine(store_only=True)
    e.load_bootstrap_yaml(default)
    assert [r[:6] for r in e.read(rtype="namespace")] == [kat_runner.parse_rel(x) for x in b["relationships"]]
    assert e.type_id("workflow") >= 0 and e.relation_id("workflow", "idempotency_key") >= 0
    # other indentation, keep / clip chomping, comments, a `---` second document with more relationships, unknown keys with bodies
    doc = ("# proxy bootstrap\nschemaFile: ignored.zed\nassertions:\n  assertTrue:\n    - \"pod:a#view@user:b\"\n\nschema: |+4\n" + indent(b["schema"], 4) +
           "\n\nrelationships: |   # seed\n      namespace:dev#viewer@user:paul\n      // a comment line of the relationship format\n\n      namespace:dev#creator@user:chani\n"
           "---\nrelationships: >-\n  pod:dev/p1#creator@user:paul\n\n  pod:dev/p1#namespace@namespace:dev\n...\n")
    e.load_bootstrap_yaml(doc)
    assert sorted(r[1] + "#" + r[2] + "@" + r[4] for r in e.read(rtype="namespace")) == ["dev#creator@chani", "dev#viewer@paul"]
    assert sorted(r[2] for r in e.read(rtype="pod")) == ["creator", "namespace"]
    # one-line scalars: plain, single- and double-quoted
    e.load_bootstrap_yaml("schema: 'definition user {}  definition doc { relation viewer: user }'\nrelationships: \"doc:d1#viewer@user:u1\\ndoc:d2#viewer@user:u2\"\n")
    assert sorted(r[1] for r in e.read(rtype="doc")) == ["d1", "d2"]
    e.load_bootstrap_yaml("schema: definition user {}\n")
    assert e.type_id("user") == 0 and e.type_id("doc") < 0
    for bad in ["relationships: |-\n  a:b#c@d:e\n",                      # no schema
                "schema: |-\n  definition user {}\nschema: |-\n  definition doc {}\n",  # duplicate key
                "schema: &anchor x\n", "schema: [a, b]\n", "  schema: x\n", "schema\n", "schema: \"unterminated\n",
                "schema:\n  multi\n  line plain\n", "schema: |-\n\tdefinition user {}\n",
                "schema: |-\n  definition user {}\nrelationships: |-\n  not a relationship\n"]:
        with pytest.raises(aclgpu.AclError) as ei:
            e.load_bootstrap_yaml(bad)
        assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT, bad
    e.close()


def test_resolve_bulk_names_to_items(aclgpu_lib):
    """acl_resolve_bulk_v (no device pass, works store-only): names -> the 16-byte items of the id entry points.  Known names get the ids acl_find
    gives, unknown object names ids without relationships (equal only when resource and subject are the same unknown object), unknown types /
    permissions and ill-formed fields a per-item error and an item no Check can answer; batches beyond 4 096 items (the interning pool) resolve
    as the single-thread path does; a resolved id of an unreferenced object sits out a fresh quarantine (Store::touch)."""
    import time
    import aclgpu
    import numpy as np
    b = kat_runner.load_bootstrap()
    e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
    rels = [("namespace", f"ns-{i}", "viewer", "user", f"user-{i % 7}", "") for i in range(40)]
    e.write([(aclgpu.OP_TOUCH, r) for r in rels])
    qs = [(rt, rid, "view", st, sid, srel) for rt, rid, _rel, st, sid, srel in rels]
    qs += [("namespace", "never-seen", "view", "user", "nobody", ""), ("namespace", "same-unknown", "view", "namespace", "same-unknown", "view"),
           ("nosuchtype", "x", "view", "user", "u", ""), ("namespace", "x", "nosuchperm", "user", "u", ""), ("namespace", "", "view", "user", "u", ""),
           ("namespace", "bad id!", "view", "user", "u", "")]
    items, err = e.resolve_bulk_views(e.make_check_views(qs))
    nt, ut = e.type_id("namespace"), e.type_id("user")
    for k, q in enumerate(qs[:len(rels)]):
        assert err[k] == 0 and items["resource_type"][k] == nt and items["permission"][k] == e.relation_id("namespace", "view")
        assert items["resource_id"][k] == e.find(q[0], q[1]) and items["subject_id"][k] == e.find(q[3], q[4]) and items["subject_type"][k] == e.type_id(q[3])
    k = len(rels)
    assert err[k] == 0 and items["resource_id"][k] >= 0xFFFFFFF0 and items["subject_id"][k] >= 0xFFFFFFF0 and items["resource_id"][k] != items["subject_id"][k]
    assert err[k + 1] == 0 and items["resource_id"][k + 1] == items["subject_id"][k + 1] >= 0xFFFFFFF0
    assert [int(x) for x in err[k + 2:]] == [aclgpu.ERR_FAILED_PRECONDITION, aclgpu.ERR_FAILED_PRECONDITION, aclgpu.ERR_INVALID_ARGUMENT, aclgpu.ERR_INVALID_ARGUMENT]
    assert all(int(t) == 0xFFFF for t in items["resource_type"][k + 2:]) and ut >= 0
    big = [qs[i % len(qs)] for i in range(9000)]  # the pool's path
    items_b, err_b = e.resolve_bulk_views(e.make_check_views(big))
    idx = np.arange(9000) % len(qs)
    assert np.array_equal(items_b, items[idx]) and np.array_equal(err_b, err[idx])
    e.close()
    # a resolved id is the caller's for a quarantine
    import os
    os.environ["ACL_ID_QUARANTINE_MS"] = "300"
    try:
        e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]), store_only=True)
    finally:
        del os.environ["ACL_ID_QUARANTINE_MS"]
    e.write([(aclgpu.OP_TOUCH, ("pod", "ns/a", "viewer", "user", "u-x", ""))])
    e.write([(aclgpu.OP_DELETE, ("pod", "ns/a", "viewer", "user", "u-x", ""))])
    ux = e.find("user", "u-x")
    time.sleep(0.4)
    items, err = e.resolve_bulk_views(e.make_check_views([("pod", "ns/a", "view", "user", "u-x", "")]))
    assert err[0] == 0 and items["subject_id"][0] == ux
    e.write([(aclgpu.OP_TOUCH, ("pod", "ns/b", "viewer", "user", "u-new", ""))])
    assert e.find("user", "u-x") == ux and e.find("user", "u-new") != ux
    e.close()
