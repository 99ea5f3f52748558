"""Pins the C oracle (oracle/acl_oracle.c) against every golden vector the
reference's tests hold for the Check/Filter path.
CPU-only."""
import pytest

from oracle import orc
from tests import kat_runner

KATS = kat_runner.load_kats()


def make_oracle(kat):
    schema, rels = kat_runner.kat_schema(kat)
    o = orc.Oracle(schema)
    if rels:
        o.write([(orc.OP_TOUCH, r) for r in rels])
    return o


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_oracle_kat(kat):
    o = make_oracle(kat)
    kat_runner.run_kat(kat, o)


@pytest.mark.parametrize("good", [
    "definition a { relation r: a | b:* }\ndefinition b {}",
    "definition u {}\ndefinition a { relation r: u\n permission p = r & r }",
    "definition u {}\ndefinition a { relation r: u\n permission p = r - r }",
    "definition u {}\ndefinition a { relation r: u | u:*\n relation q: a\n permission p = (r - q->p) & r + nil }",
    "definition u {}\ndefinition a { relation r: a\n relation v: u\n permission p = v + r.all(p) - r.any(p) }",  # the intersection arrow
])
def test_oracle_accepts_round4_schema_features(good):
    orc.Oracle(good).close()


@pytest.mark.parametrize("bad", [
    "definition a { relation r: a | b:x }\ndefinition b {}",
    "definition u {}\ndefinition a { relation r: u | u:*\n relation q: a\n permission p = r->q }",  # arrow over a relation that allows wildcards
    "definition u {}\ndefinition a { relation r: u\n permission p = r & }",
    "caveat c(x int) { x > 1 }\ndefinition u {}",
    "definition u {}\ndefinition a { relation r: u with c }",
    "definition u {}\ndefinition a { relation r: a\n permission p = r.some(p) }",  # (only .any() and .all() are arrow functions)
    "definition u {}\ndefinition b {}\ndefinition a { relation r: a | b\n relation v: u\n permission p = v + r.all(p) }",  # .all() over a subject type without the permission: refused (fails closed)
    "definition a { relation r: nosuch }",
    "definition u {}\ndefinition a { relation r: u\n permission p = nosuch }",
])
def test_oracle_rejects_unsupported_schema(bad):
    with pytest.raises(orc.OracleError):
        orc.Oracle(bad)


def test_oracle_depth_limit_chain():
    """Dispatch depth 50 (pkg/spicedb/spicedb.go:34): a chain of nested groups
    answers HAS while the number of dispatches is <= 50 and errors beyond."""
    schema = "definition user {}\ndefinition group { relation member: user | group#member }"
    o = orc.Oracle(schema)
    n = 60
    o.write([(orc.OP_TOUCH, f"group:g{i}#member@group:g{i+1}#member") for i in range(n)])
    o.write([(orc.OP_TOUCH, f"group:g{n}#member@user:deep")])
    # check from gK: dispatches needed = (n-K)+1  (gK, gK+1, ..., gn)
    for k, want in [(n, 2), (n - 10, 2), (n - 49, 2), (n - 50, 0), (0, 0)]:
        perm, err = o.check("group", f"g{k}", "member", "user", "deep")
        assert perm == want, (k, perm, err)
        assert (err == orc.ERR_DEPTH) == (want == 0)
