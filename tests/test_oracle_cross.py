"""Cross-checks the two independent oracle restatements (C: oracle/acl_oracle.c,
Python: oracle/pyoracle.py) on hypothesis-generated graphs, including cyclic
group nesting -- the shapes no reference test pins (SURVEY.md 8(c) last row)."""
import os

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import orc
from oracle.pyoracle import PyOracle
from tests.outcomes import outcome

SCHEMA = """
definition user {}
definition group {
  relation member: user | group#member
  relation owner: user
  permission manage = owner + member
}
definition org {
  relation admin: user | group#member
  relation parent: org
  permission view = admin + parent->view
}
definition doc {
  relation org: org
  relation viewer: user | group#member | group#manage
  relation creator: user
  permission edit = creator
  permission view = viewer + edit + org->view
  permission nothing = nil
}
"""

USERS = [f"u{i}" for i in range(4)]
GROUPS = [f"g{i}" for i in range(4)]
ORGS = [f"o{i}" for i in range(3)]
DOCS = [f"d{i}" for i in range(4)]


def tuples_strategy():
    user = st.sampled_from(USERS)
    group = st.sampled_from(GROUPS)
    org = st.sampled_from(ORGS)
    doc = st.sampled_from(DOCS)
    one = st.one_of(
        st.tuples(st.just("group"), group, st.just("member"), st.just("user"), user, st.just("")),
        st.tuples(st.just("group"), group, st.just("member"), st.just("group"), group, st.just("member")),
        st.tuples(st.just("group"), group, st.just("owner"), st.just("user"), user, st.just("")),
        st.tuples(st.just("org"), org, st.just("admin"), st.just("user"), user, st.just("")),
        st.tuples(st.just("org"), org, st.just("admin"), st.just("group"), group, st.just("member")),
        st.tuples(st.just("org"), org, st.just("parent"), st.just("org"), org, st.just("")),
        st.tuples(st.just("doc"), doc, st.just("org"), st.just("org"), org, st.just("")),
        st.tuples(st.just("doc"), doc, st.just("viewer"), st.just("user"), user, st.just("")),
        st.tuples(st.just("doc"), doc, st.just("viewer"), st.just("group"), group, st.just("member")),
        st.tuples(st.just("doc"), doc, st.just("viewer"), st.just("group"), group, st.just("manage")),
        st.tuples(st.just("doc"), doc, st.just("creator"), st.just("user"), user, st.just("")),
    )
    return st.lists(one, min_size=0, max_size=24)


def all_queries():
    qs = []
    subjects = [("user", u, "") for u in USERS] + [("group", GROUPS[0], "member"), ("group", GROUPS[1], "manage")]
    for s in subjects:
        for d in DOCS:
            for p in ("view", "edit", "viewer", "nothing"):
                qs.append(("doc", d, p) + s)
        for o in ORGS:
            qs.append(("org", o, "view") + s)
        for g in GROUPS:
            for p in ("member", "manage"):
                qs.append(("group", g, p) + s)
    return qs


QUERIES = all_queries()
PY2C = {"HAS": (orc.PERM_HAS, 0), "NO": (orc.PERM_NO, 0), "ERR": (orc.PERM_UNSPEC, orc.ERR_DEPTH)}


@settings(max_examples=60, deadline=None)
@given(tuples_strategy())
def test_c_oracle_matches_python_oracle(tuples):
    co = orc.Oracle(SCHEMA)
    po = PyOracle(SCHEMA)
    if tuples:
        co.write([(orc.OP_TOUCH, t) for t in dict.fromkeys(tuples)])
    for t in tuples:
        po.touch(*t)
    for q in QUERIES:
        assert co.check(*q) == PY2C[po.check(*q)], q
    for s in [("user", USERS[0], ""), ("group", GROUPS[0], "member")]:
        for rt, p in [("doc", "view"), ("org", "view"), ("group", "member"), ("group", "manage")]:
            assert outcome(co.lookup, rt, p, *s) == outcome(po.lookup_resources, rt, p, *s), (rt, p, s)
    # the TUNED CPU evaluator (bench.py's cpu_baseline.tuned: row index, level-synchronous frontier, merged states) answers every query as the recursive
    # one does -- cycles, usersets that are their own members, chains across the depth limit included
    assert co.tuned_build()
    for q in QUERIES:
        rt, rid, perm, st, sid, srel = q
        got = co.tuned_check_bulk_ids_mt(2, rt, perm, [co.intern(rt, rid)], st, srel, [co.intern(st, sid)])
        assert (int(got[0][0]), int(got[1][0])) == co.check(*q), q


def test_tuned_cpu_check_equals_the_recursive_oracle_on_the_workloads():
    """... and on the BASELINE-shaped graphs, whole batches, several threads (dynamic chunks); a schema with `&` / `-` is refused, not guessed at."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spicedb-kubeapi-proxy_amd"))
    from aclgpu import workloads
    for w in (workloads.c1(), workloads.c2(scale=0.05, batch=6000), workloads.c4(scale=0.02, batch=20000, n_user=20000)):
        o = orc.Oracle(w.schema)
        w.load(o)
        o.freeze()
        rt, perm, st = w.check
        want = o.check_bulk_ids_mt(4, rt, perm, w.res, st, "", w.subj)
        got = o.tuned_check_bulk_ids_mt(5, rt, perm, w.res, st, "", w.subj)
        assert got is not None and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), w.name
    # a 60-long nesting chain: HAS up to the depth limit, the depth error beyond it, exactly where the recursive evaluator puts them
    o = orc.Oracle(SCHEMA)
    o.write([(orc.OP_TOUCH, ("group", "c0", "member", "user", "deep", ""))] + [(orc.OP_TOUCH, ("group", f"c{i + 1}", "member", "group", f"c{i}", "member")) for i in range(60)])
    ids = [o.intern("group", f"c{i}") for i in range(61)]
    want = o.check_bulk_ids("group", "member", ids, "user", "", [o.intern("user", "deep")] * 61)
    got = o.tuned_check_bulk_ids_mt(3, "group", "member", ids, "user", "", [o.intern("user", "deep")] * 61)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and (want[0] == 2).sum() == 50 and (want[1] == orc.ERR_DEPTH).sum() == 11
    nm = orc.Oracle(SCHEMA_NM)
    assert not nm.tuned_build() and nm.tuned_check_bulk_ids_mt(2, "doc", "view", [0], "user", "", [0]) is None


def test_bytes_model_result_agrees_with_check():
    """orc_check_bytes (level-synchronous byte model) must reach the
    same allow/deny as the recursive evaluation on acyclic data."""
    co = orc.Oracle(SCHEMA)
    rels = [("group", "g0", "member", "user", "u0", ""), ("group", "g1", "member", "group", "g0", "member"),
            ("org", "o0", "admin", "group", "g1", "member"), ("org", "o1", "parent", "org", "o0", ""),
            ("doc", "d0", "org", "org", "o1", ""), ("doc", "d1", "viewer", "user", "u1", "")]
    co.write([(orc.OP_TOUCH, r) for r in rels])
    L = co._L
    tid = co.type_id
    for d, u, want in [("d0", "u0", 2), ("d0", "u1", 1), ("d1", "u1", 2), ("d1", "u0", 1)]:
        # numeric ids are the interned ids in text mode: recover via a lookup of names
        did = [i for i in range(8) if L.orc_object_name(co._h, tid("doc"), i) == d.encode()][0]
        uid = [i for i in range(8) if L.orc_object_name(co._h, tid("user"), i) == u.encode()][0]
        b, r = co.check_bytes("doc", "view", did, "user", "", uid)
        assert r == want and b > 17


# ---- round 4: intersection, exclusion, wildcards.  Group nesting may be cyclic, so depth errors meet `&` and
# `-` from both sides; arrows reach permissions that are themselves non-monotone; wildcards sit in positive and in subtracted operands.
SCHEMA_NM = """
definition user {}
definition group {
  relation member: user | group#member | user:*
  relation banned: user | group#member
  permission active = member - banned
}
definition folder {
  relation parent: folder
  relation viewer: user | group#member | group#active | user:*
  relation banned: user | user:*
  relation auditor: user | group#member
  permission view = (viewer - banned) + parent->view
  permission audit = auditor & view
  permission sealed = parent.all(view) - banned
}
definition doc {
  relation folder: folder
  relation viewer: user | group#active
  relation editor: user | group#member
  relation banned: user | group#member
  permission edit = editor - banned
  permission view = viewer + edit + folder->view - banned
  permission strict = viewer & editor & folder->audit
  permission odd = (viewer - (editor & folder->view)) + (edit & folder->view)
  permission nothing = nil & viewer
  permission everywhere = folder.all(view)
  permission vetted = viewer + folder.all(audit) & folder.any(view)
  permission deep_all = folder.all(sealed)
}
"""
FOLDERS = [f"f{i}" for i in range(3)]


def nm_tuples_strategy():
    user = st.sampled_from(USERS + ["*"])
    plain = st.sampled_from(USERS)
    group = st.sampled_from(GROUPS)
    folder = st.sampled_from(FOLDERS)
    doc = st.sampled_from(DOCS)
    j = st.just
    one = st.one_of(
        st.tuples(j("group"), group, j("member"), j("user"), user, j("")),
        st.tuples(j("group"), group, j("member"), j("group"), group, j("member")),
        st.tuples(j("group"), group, j("banned"), j("user"), plain, j("")),
        st.tuples(j("group"), group, j("banned"), j("group"), group, j("member")),
        st.tuples(j("folder"), folder, j("parent"), j("folder"), folder, j("")),
        st.tuples(j("folder"), folder, j("viewer"), j("user"), user, j("")),
        st.tuples(j("folder"), folder, j("viewer"), j("group"), group, st.sampled_from(["member", "active"])),
        st.tuples(j("folder"), folder, j("banned"), j("user"), user, j("")),
        st.tuples(j("folder"), folder, j("auditor"), j("user"), plain, j("")),
        st.tuples(j("folder"), folder, j("auditor"), j("group"), group, j("member")),
        st.tuples(j("doc"), doc, j("folder"), j("folder"), folder, j("")),
        st.tuples(j("doc"), doc, j("viewer"), j("user"), plain, j("")),
        st.tuples(j("doc"), doc, j("viewer"), j("group"), group, j("active")),
        st.tuples(j("doc"), doc, j("editor"), j("user"), plain, j("")),
        st.tuples(j("doc"), doc, j("editor"), j("group"), group, j("member")),
        st.tuples(j("doc"), doc, j("banned"), j("user"), plain, j("")),
        st.tuples(j("doc"), doc, j("banned"), j("group"), group, j("member")),
    )
    return st.lists(one, min_size=0, max_size=30)


def nm_queries():
    qs = []
    subjects = [("user", u, "") for u in USERS] + [("group", GROUPS[0], "member"), ("group", GROUPS[1], "active")]
    for s in subjects:
        for d in DOCS:
            for p in ("view", "edit", "strict", "odd", "nothing", "viewer", "everywhere", "vetted", "deep_all"):
                qs.append(("doc", d, p) + s)
        for f in FOLDERS:
            for p in ("view", "audit", "sealed"):
                qs.append(("folder", f, p) + s)
        for g in GROUPS:
            for p in ("member", "active"):
                qs.append(("group", g, p) + s)
    return qs


NM_QUERIES = nm_queries()


@settings(max_examples=200, deadline=None)
@given(nm_tuples_strategy())
def test_c_oracle_matches_python_oracle_nonmonotone(tuples):
    co = orc.Oracle(SCHEMA_NM)
    po = PyOracle(SCHEMA_NM)
    if tuples:
        co.write([(orc.OP_TOUCH, t) for t in dict.fromkeys(tuples)])
    for t in tuples:
        po.touch(*t)
    for q in NM_QUERIES:
        assert co.check(*q) == PY2C[po.check(*q)], q
    for s in [("user", USERS[0], ""), ("group", GROUPS[0], "member")]:
        for rt, p in [("doc", "view"), ("doc", "odd"), ("doc", "strict"), ("folder", "audit"), ("group", "active"), ("doc", "everywhere"), ("doc", "vetted"),
                      ("folder", "sealed")]:
            assert outcome(co.lookup, rt, p, *s) == outcome(po.lookup_resources, rt, p, *s), (rt, p, s)


def test_precedence_and_three_valued_rules():
    """The restated operator precedence (`-` loosest, then `&`, then `+`) and the fixed order of the three-valued rules, on both oracles."""
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
    }
    """
    co, po = orc.Oracle(schema), PyOracle(schema)
    rels = [("d", "x", "a", "user", "u", ""), ("d", "x", "c", "user", "u", ""), ("d", "y", "a", "user", "u", ""), ("d", "y", "b", "user", "u", ""),
            # a 60-long membership chain: g0 <- g1 <- ... (checking g59#member for a user in g0 crosses the depth limit)
            ("g", "g0", "member", "user", "deep", "")] + [("g", f"g{i + 1}", "member", "g", f"g{i}", "member") for i in range(60)] + [
            ("d", "e1", "a", "user", "deep", ""), ("d", "e1", "b", "g", "g59", "member"),   # base HAS, subtracted ERR -> ERR ; a & b: HAS & ERR -> ERR
            ("d", "e2", "b", "g", "g59", "member"),                                          # base NO: the subtracted error is never looked at -> NO ; NO & ERR -> NO
            ("d", "e3", "a", "g", "g59", "member"), ("d", "e3", "b", "user", "deep", "")]    # base ERR -> ERR ; ERR & HAS -> ERR
    co.write([(orc.OP_TOUCH, r) for r in rels])
    for r in rels:
        po.touch(*r)
    H, N, E = (orc.PERM_HAS, 0), (orc.PERM_NO, 0), (orc.PERM_UNSPEC, orc.ERR_DEPTH)
    want = {("x", "p1"): N,   # (a + b) - c
            ("x", "p2"): N,   # a - (b + c)
            ("x", "p3"): H,   # a & (b + c)
            ("x", "p4"): H,   # a - (b & c): b is empty
            ("y", "p4"): H, ("y", "p5"): N, ("x", "p5"): N, ("y", "p1"): H, ("y", "p2"): N, ("y", "p3"): H}
    for (obj, perm), w in want.items():
        assert co.check("d", obj, perm, "user", "u") == w == PY2C[po.check("d", obj, perm, "user", "u")], (obj, perm)
    for obj, perm, w in [("e1", "deny_err", E), ("e1", "and_err", E), ("e2", "deny_err", N), ("e2", "and_err", N), ("e3", "deny_err", E), ("e3", "and_err", E)]:
        assert co.check("d", obj, perm, "user", "deep") == w == PY2C[po.check("d", obj, perm, "user", "deep")], (obj, perm)
    # LookupResources over these permissions: e1 is a CANDIDATE of `deep` (the positive operand `a` names it) and its Check errs -- the reference's stream
    # ends at that error and the list request fails (pkg/authz/lookups.go:75-83, responsefilterer.go:196-204), so both restatements fail the call with
    # the item's code; e3 is no candidate for `deny_err` (its `a` lies beyond the depth limit: no reverse walk reaches it) but is one for `and_err` (`b`)
    for perm in ("deny_err", "and_err"):
        assert outcome(co.lookup, "d", perm, "user", "deep") == outcome(po.lookup_resources, "d", perm, "user", "deep") == ("err", orc.ERR_DEPTH), perm
    assert outcome(co.lookup, "d", "p3", "user", "u") == outcome(po.lookup_resources, "d", "p3", "user", "u") == ("ok", {"x", "y"})
    # the lenient form (the engine's ACL_FLAG_LENIENT_LOOKUP): the erring candidates are dropped, the call succeeds
    co.set_lenient_lookup(True)
    po.lenient_lookup = True
    for perm in ("deny_err", "and_err"):
        assert co.lookup("d", perm, "user", "deep") == po.lookup_resources("d", perm, "user", "deep") == set(), perm
    assert co.lookup("d", "p1", "user", "deep") == po.lookup_resources("d", "p1", "user", "deep") == {"e1", "e3"}  # (c is empty: nothing subtracted, HAS beats the error under `+`)


def test_cycles_through_nonmonotone_permissions_agree():
    """Branching cycles under `-` and `&`, and a cycle THROUGH a non-monotone permission (team#active members of teams): the recursive Python
    oracle (no memo: every path walked to the depth limit... on a graph this small) and the memoising C oracle give the same answers and the
    same lookup sets -- the vectors tests/test_combine_gpu.py::test_branching_cycles_under_exclusion_and_intersection holds the engine to."""
    import re
    import os
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_combine_gpu.py")).read()
    blk = src[src.index("def test_branching_cycles_under_exclusion_and_intersection"):]
    schema = re.search(r'schema = """(.*?)"""', blk, re.S).group(1)
    rels = eval("[" + re.search(r"rels = \[(.*?)\]\n    o = ", blk, re.S).group(1) + "]")
    co, po = orc.Oracle(schema), PyOracle(schema)
    co.write([(orc.OP_TOUCH, r) for r in rels])
    for r in rels:
        m = re.match(r"(\w+):([^#]+)#(\w+)@(\w+):([^#]+)(?:#(\w+))?$", r)
        po.touch(m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), m.group(6) or "")
    qs = [("group", g, p, "user", u, "") for g in ("g0", "g1", "g2") for p in ("active", "inner", "member") for u in ("deep", "outcast", "nobody")]
    qs += [("team", t, "active", "user", u, "") for t in ("t0", "t1") for u in ("deep", "outcast", "nobody")]
    for q in qs:
        assert co.check(*q) == PY2C[po.check(*q)], q
    for rt, perm in (("group", "active"), ("group", "inner")):
        for u in ("deep", "outcast"):
            assert outcome(co.lookup, rt, perm, "user", u, "") == outcome(po.lookup_resources, rt, perm, "user", u, ""), (rt, perm, u)


def test_a_depth_error_does_not_depend_on_the_subject():
    """What the engine's depth sweep rests on (csrc/engine.cpp no_object_is_deep): a Check that does not find its subject explores every path below its resource,
    so whether it ends at the dispatch-depth limit (spicedb.go:34) is a property of the RESOURCE -- the same for a subject nobody is.  Random group graphs with
    cycles, self-memberships and chains across the limit, unions and arrows only (the route is not taken for `&` / `-`), both oracles."""
    import random
    schema = """definition user {}
definition group { relation member: user | group#member
 relation parent: group
 permission reach = member + parent->reach }
definition pod { relation viewer: user | group#member | user:*
 relation owner: group
 permission view = viewer + owner->reach }"""
    rng = random.Random(0x5ACE0D)
    for trial in range(12):
        ng, nu, npod = 14, 6, 20
        rels = set()
        for g in range(ng):
            for _ in range(rng.randrange(0, 3)):
                rels.add(("group", f"g{g}", "member", "user", f"u{rng.randrange(nu)}", ""))
            for _ in range(rng.randrange(0, 3)):
                rels.add(("group", f"g{g}", "member", "group", f"g{rng.randrange(ng)}", "member"))  # (cycles and g#member @ g#member included)
            if rng.random() < 0.3:
                rels.add(("group", f"g{g}", "parent", "group", f"g{rng.randrange(ng)}", ""))
        if trial % 3 == 0:  # a chain of 60 nested groups: deeper than the limit without a cycle
            for k in range(60):
                rels.add(("group", f"c{k}", "member", "group", f"c{k + 1}", "member"))
            rels.add(("group", "c60", "member", "user", "u0", ""))
            rels.add(("pod", "p0", "viewer", "group", "c0", "member"))
            rels.add(("pod", "p1", "viewer", "group", "c30", "member"))
        for p in range(npod):
            for _ in range(rng.randrange(0, 3)):
                rels.add(("pod", f"p{p}", "viewer", "group", f"g{rng.randrange(ng)}", "member"))
            if rng.random() < 0.3:
                rels.add(("pod", f"p{p}", "viewer", "user", f"u{rng.randrange(nu)}", ""))
            if rng.random() < 0.3:
                rels.add(("pod", f"p{p}", "owner", "group", f"g{rng.randrange(ng)}", ""))
            if rng.random() < 0.05:
                rels.add(("pod", f"p{p}", "viewer", "user", "*", ""))
        co, po = orc.Oracle(schema), PyOracle(schema)
        co.write([(orc.OP_TOUCH, f"{a}:{b}#{c}@{d}:{e}" + (f"#{f}" if f else "")) for a, b, c, d, e, f in sorted(rels)])
        for t in sorted(rels):
            po.touch(*t)
        seen = set()
        for rt, perm, ids in (("pod", "view", [f"p{p}" for p in range(npod)]), ("group", "reach", [f"g{g}" for g in range(ng)]), ("group", "member", ["c0", "c20"])):
            for rid in ids:
                nobody = co.check(rt, rid, perm, "user", "nobody-at-all", "")
                assert nobody == PY2C[po.check(rt, rid, perm, "user", "nobody-at-all", "")]
                for u in range(nu):
                    got = co.check(rt, rid, perm, "user", f"u{u}", "")
                    seen.add(got)
                    if got[0] != orc.PERM_HAS:
                        assert got == nobody, (trial, rt, rid, perm, u, got, nobody)  # NO for both or the depth error for both
                    if nobody[0] == orc.PERM_HAS:  # (a wildcard: then everybody has it)
                        assert got == nobody
        assert (orc.PERM_HAS, 0) in seen and (orc.PERM_NO, 0) in seen and (trial % 3 or (orc.PERM_UNSPEC, orc.ERR_DEPTH) in seen)
