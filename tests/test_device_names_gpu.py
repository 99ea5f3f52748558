"""String calls of PostFilter size can resolve their object names on the device (ACL_DEVICE_NAMES=1; csrc/engine_names.cpp, kernels.hip
k_resolve_names; reference call site pkg/authz/postfilter.go:67-134).  The bar is the host path's: identical permissionship and error code for
every item -- against the CPU oracle, and against the same engine with the names resolved by the host -- with acl_stats().device_name_calls
proving which path answered."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu

SCHEMA = """definition user {}
definition group { relation member: user | group#member }
definition namespace { relation viewer: user | group#member
 relation creator: user
 permission view = viewer + creator }
definition pod { relation namespace: namespace
 relation creator: user
 permission view = creator + namespace->view }"""


@pytest.fixture(scope="module")
def aclgpu(aclgpu_lib):
    import aclgpu as m
    return m


def name_of_length(k, salt):
    """an object id of exactly k bytes (1..46: what a name slot holds inline), distinct per (k, salt)"""
    s = f"{salt}-{k}-"
    return (s + "x" * k)[:k] if k > len(s) else "abcdefghijklmnopqrstuvwxyz0123456789/_|-=+"[(k * 7 + salt) % 42] * k


def both_forms(e, qs):
    return [list(zip(*[x.tolist() for x in call(prep)])) for prep, call in ((e.make_check_views(qs), e.check_bulk_views), (e.make_check_strings_named(qs), e.check_bulk_prepared))]


def test_names_of_every_length_known_and_unknown(aclgpu, monkeypatch):
    """ids of 1..46 bytes (every tail length of the hash, every mask of the compare), known and unknown, a subject that is its own unknown
    resource, items that carry an error in the middle of the batch, `...` as subject relation -- device path = host path = oracle."""
    monkeypatch.setenv("ACL_DEVICE_NAMES", "1")
    monkeypatch.setenv("ACL_DEVICE_NAMES_MIN", "64")
    rels = []
    for k in range(1, 47):
        rels.append(f"namespace:{name_of_length(k, 1)}#viewer@user:{name_of_length(k, 2)}")
        rels.append(f"pod:{name_of_length(k, 3)}#namespace@namespace:{name_of_length(k, 1)}")
        rels.append(f"pod:{name_of_length(k, 3)}#creator@user:{name_of_length((k * 5) % 46 + 1, 4)}")
    rels = sorted(set(rels))
    o = orc.Oracle(SCHEMA)
    o.write([(orc.OP_TOUCH, r) for r in rels])
    rng = np.random.default_rng(11)
    qs = []
    for _ in range(3000):
        k = int(rng.integers(1, 47))
        j = int(rng.integers(1, min(46, 52 - ((k + 3) & ~3)) + 1))  # (both ids, each padded to whole dwords, share the 52 name bytes of a record)
        kind = int(rng.integers(0, 6))
        kk = (k - 1) % 24 + 1  # (a pair of ids of the same length: two times at most 24 bytes)
        if kind == 0:
            qs.append(("pod", name_of_length(kk, 3), "view", "user", name_of_length(kk, 2), ""))        # granted through the namespace
        elif kind == 1:
            qs.append(("pod", name_of_length(k, 3), "view", "user", name_of_length(j, 4), "..."))       # sometimes the creator
        elif kind == 2:
            qs.append(("pod", name_of_length(k, 9), "view", "user", name_of_length(j, 2), ""))          # unknown resource
        elif kind == 3:
            qs.append(("namespace", name_of_length(k, 1), "view", "user", name_of_length(j, 8), ""))    # unknown subject
        elif kind == 4:
            qs.append(("pod", name_of_length(kk, 7), "view", "pod", name_of_length(kk, 7), ""))         # the same unknown object twice
        else:
            qs.append(("namespace", name_of_length(kk, 1), "view", "user", name_of_length(kk, 2), ""))
    qs[7] = ("nosuchtype", "x", "view", "user", "u1", "")
    qs[8] = ("pod", name_of_length(5, 3), "nosuchperm", "user", "u1", "")
    qs[1500] = ("pod", name_of_length(5, 3), "view", "user", "u1", "nosuchrel")
    qs[2999] = ("pod", name_of_length(46, 3), "view", "user", name_of_length(4, 4), "")
    qs[2998] = ("pod", name_of_length(4, 3), "view", "user", name_of_length(46, 4), "")
    want = [o.check(*q) for q in qs]
    assert {w[1] for w in want} >= {0, aclgpu.ERR_FAILED_PRECONDITION} and 100 < sum(w[0] == 2 for w in want) < len(want) - 100
    with aclgpu.Engine(SCHEMA, "\n".join(rels)) as e:
        for got in both_forms(e, qs):
            assert got == want
        assert e.stats()["device_name_calls"] == 2
        # a name longer than a slot holds inline, or two names that do not fit one record: the whole call goes the host's way, same answers
        long_qs = qs[:200] + [("pod", "p" * 47, "view", "user", "u1", ""), ("pod", "a" * 30, "view", "user", "b" * 30, "")] + qs[200:400]
        want_long = want[:200] + [o.check(*long_qs[200]), o.check(*long_qs[201])] + want[200:400]
        for got in both_forms(e, long_qs):
            assert got == want_long
        assert e.stats()["device_name_calls"] == 2
        # requests the API's validation refuses fail as a whole on either path (an ill-formed UNKNOWN id is only seen after the device looked)
        for bad in (("pod", "", "view", "user", "u1", ""), ("pod", "has.dot", "view", "user", "u1", ""), ("pod", name_of_length(3, 3), "view", "user", "*", ""),
                    ("pod", name_of_length(3, 3), "view", "user", "system:admin", "")):
            with pytest.raises(aclgpu.AclError) as ei:
                e.check_bulk(qs[:300] + [bad] + qs[300:600])
            assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT, bad
    monkeypatch.delenv("ACL_DEVICE_NAMES")  # (the default: names resolved by the host's interning threads)
    with aclgpu.Engine(SCHEMA, "\n".join(rels)) as e_host:
        for got in both_forms(e_host, qs):
            assert got == want
        assert e_host.stats()["device_name_calls"] == 0
    monkeypatch.setenv("ACL_DEVICE_NAMES", "1")
    with aclgpu.Engine(SCHEMA, "\n".join(rels), per_item_validation=True) as e_lax:  # the ill-formed item fails ITS pair
        bad = ("pod", "has.dot", "view", "user", "u1", "")
        got = e_lax.check_bulk(qs[:300] + [bad] + qs[300:600])
        pairs = list(zip(*got))
        assert pairs[300] == (0, aclgpu.ERR_INVALID_ARGUMENT) and pairs[:300] == want[:300] and pairs[301:] == want[300:600]
        assert e_lax.stats()["device_name_calls"] == 1


def test_the_copy_follows_the_tables(aclgpu, monkeypatch):
    """Writes between string calls: new names (single slots written in place), thousands of them (the table re-hashes: a new array), objects
    that lose their last relationship and whose ids go to new names (quarantine 0: a tombstone and a new slot) -- after every step the device
    path answers as the oracle does, for the old names, the new ones and the ones that left."""
    monkeypatch.setenv("ACL_DEVICE_NAMES", "1")
    monkeypatch.setenv("ACL_DEVICE_NAMES_MIN", "64")
    monkeypatch.setenv("ACL_ID_QUARANTINE_MS", "0")
    rels = [f"pod:ns{i % 5}/p{i}#creator@user:u{i % 17}" for i in range(200)]
    o = orc.Oracle(SCHEMA)
    o.write([(orc.OP_TOUCH, r) for r in rels])

    def ask(e, pods, users, calls_before):
        qs = [("pod", p, "view", "user", u, "") for p in pods for u in users]
        want = [o.check(*q) for q in qs]
        got = list(zip(*[x.tolist() for x in e.check_bulk_views(e.make_check_views(qs))]))
        assert got == want
        assert e.stats()["device_name_calls"] == calls_before + 1
        return sum(w[0] == 2 for w in want)

    with aclgpu.Engine(SCHEMA, "\n".join(rels)) as e:
        users = [f"u{i}" for i in range(17)] + ["nobody"]
        calls = 0
        assert ask(e, [f"ns{i % 5}/p{i}" for i in range(0, 200, 3)], users, calls) > 0
        calls += 1
        # a few new names: slots written in place
        ups = [(aclgpu.OP_TOUCH, ("pod", f"fresh/p{i}", "creator", "user", f"newuser{i}", "")) for i in range(20)]
        e.write(ups)
        o.write(ups)
        assert ask(e, [f"fresh/p{i}" for i in range(20)] + ["ns0/p0"], [f"newuser{i}" for i in range(20)] + ["u0"], calls) >= 20
        calls += 1
        # thousands of new names: the pod and user tables re-hash more than once
        for b in range(0, 6000, 1000):
            ups = [(aclgpu.OP_TOUCH, ("pod", f"bulk{b}/p{i}", "creator", "user", f"bulkuser{i % 700}", "")) for i in range(b, b + 1000)]
            e.write(ups)
            o.write(ups)
        assert ask(e, [f"bulk{(i // 1000) * 1000}/p{i}" for i in range(0, 6000, 61)] + ["fresh/p3"], [f"bulkuser{i}" for i in range(0, 700, 23)] + ["newuser3"], calls) > 0
        calls += 1
        # objects leave, their ids are handed to new names at once (no quarantine): the old names must stop resolving, the new ones resolve
        dels = [(aclgpu.OP_DELETE, ("pod", f"fresh/p{i}", "creator", "user", f"newuser{i}", "")) for i in range(20)]
        e.write(dels)
        o.write(dels)
        ups = [(aclgpu.OP_TOUCH, ("pod", f"later/p{i}", "creator", "user", f"lateuser{i}", "")) for i in range(20)]
        e.write(ups)
        o.write(ups)
        assert e.stats()["ids_recycled"] > 0
        n_has = ask(e, [f"fresh/p{i}" for i in range(20)] + [f"later/p{i}" for i in range(20)], [f"newuser{i}" for i in range(20)] + [f"lateuser{i}" for i in range(20)], calls)
        assert n_has == 20
