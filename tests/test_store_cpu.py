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


@pytest.mark.parametrize("bad", [
    "definition a { relation r: a | b:* }\ndefinition b {}",
    "definition u {}\ndefinition a { relation r: u\n permission p = r & r }",
    "definition u {}\ndefinition a { relation r: u\n permission p = r - r }",
    "caveat c(x int) { x > 1 }\ndefinition u {}",
    "definition u {}\ndefinition a { relation r: u with c }",
    "definition u {}\ndefinition a { relation r: a\n permission p = r.all(p) }",
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
