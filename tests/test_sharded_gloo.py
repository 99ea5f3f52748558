"""world_size-2 (and 3) gloo tests of the sharded-graph protocol (aclgpu/sharded.py), on CPU.

The N>1 path of SURVEY.md 8(e): per-level export -> all-gather -> import -> reduce.  The product's protocol
code runs unchanged; the shard itself is the CpuShard test double (tests/shard_double.py) because the HIP
engine has no CPU path.  Expected answers come from the recursive Python oracle on the UNSHARDED graph."""
import json
import os
import random
import socket
import subprocess
import sys

import pytest

from oracle.pyoracle import PyOracle
from tests.test_oracle_cross import SCHEMA, USERS, GROUPS, ORGS, DOCS, QUERIES

HERE = os.path.dirname(os.path.abspath(__file__))
P2C = {"HAS": (2, 0), "NO": (1, 0), "ERR": (0, 100)}


def random_tuples(rng, n):
    kinds = [
        lambda: ("group", rng.choice(GROUPS), "member", "user", rng.choice(USERS), ""),
        lambda: ("group", rng.choice(GROUPS), "member", "group", rng.choice(GROUPS), "member"),
        lambda: ("group", rng.choice(GROUPS), "owner", "user", rng.choice(USERS), ""),
        lambda: ("org", rng.choice(ORGS), "admin", "user", rng.choice(USERS), ""),
        lambda: ("org", rng.choice(ORGS), "admin", "group", rng.choice(GROUPS), "member"),
        lambda: ("org", rng.choice(ORGS), "parent", "org", rng.choice(ORGS), ""),
        lambda: ("doc", rng.choice(DOCS), "org", "org", rng.choice(ORGS), ""),
        lambda: ("doc", rng.choice(DOCS), "viewer", "user", rng.choice(USERS), ""),
        lambda: ("doc", rng.choice(DOCS), "viewer", "group", rng.choice(GROUPS), "member"),
        lambda: ("doc", rng.choice(DOCS), "viewer", "group", rng.choice(GROUPS), "manage"),
        lambda: ("doc", rng.choice(DOCS), "creator", "user", rng.choice(USERS), ""),
    ]
    return list(dict.fromkeys(rng.choice(kinds)() for _ in range(n)))


CHAIN_SCHEMA = """
definition user {}
definition team { relation member: user | crew#member }
definition crew { relation member: user | team#member }
"""


def chain_case(n):
    """team:t0 <- crew:c0 <- team:t1 <- ... alternating types (every hop crosses shards when they are split):
    the user sits n userset hops below team:t0#member."""
    tuples = []
    for i in range(n):
        a = ("team", f"t{i // 2}") if i % 2 == 0 else ("crew", f"c{i // 2}")
        b = ("crew", f"c{i // 2}") if i % 2 == 0 else ("team", f"t{i // 2 + 1}")
        tuples.append((a[0], a[1], "member", b[0], b[1], "member"))
    last = ("crew", f"c{(n - 1) // 2}") if (n - 1) % 2 == 0 else ("team", f"t{(n - 1) // 2 + 1}")
    tuples.append((last[0], last[1], "member", "user", "deep", ""))
    return tuples


def make_cases():
    rng = random.Random(0x5ACE0E)
    cases = []
    lookups = [("doc", "view", "user", USERS[0], ""), ("org", "view", "user", USERS[1], ""), ("group", "member", "group", GROUPS[0], "member"),
               ("doc", "view", "group", GROUPS[0], "member"), ("group", "manage", "user", USERS[2], "")]
    for n in (0, 6, 14, 24, 40):
        cases.append({"schema": SCHEMA, "tuples": random_tuples(rng, n), "queries": QUERIES, "lookups": lookups})
    # tiny export buffer: the first attempt drops entries, every rank must agree to grow and redo the batch
    cases.append({"schema": SCHEMA, "tuples": random_tuples(rng, 40), "queries": QUERIES, "lookups": lookups, "export_entries": 1})
    # dispatch-depth limit across shards: 49 hops below -> HAS at exactly 50 dispatches; 50 hops -> depth error
    for hops in (49, 50):
        cases.append({"schema": CHAIN_SCHEMA, "tuples": chain_case(hops), "lookups": [("team", "member", "user", "deep", "")],
                      "queries": [("team", "t0", "member", "user", "deep", ""), ("team", "t0", "member", "user", "nobody", ""),
                                  ("crew", "c0", "member", "user", "deep", "")]})
    return cases


def run_world(world, cases, tmp_path, exchange="allgather"):
    cf, of = tmp_path / "cases.json", tmp_path / "out.json"
    cf.write_text(json.dumps(cases))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", ACL_EXCHANGE=exchange)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), str(cf), str(of)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return json.loads(of.read_text())


@pytest.mark.parametrize("world,exchange", [(2, "allgather"), (3, "allgather"), (2, "alltoall"), (3, "alltoall")])
def test_sharded_protocol_matches_unsharded_oracle(world, exchange, tmp_path):
    cases = make_cases()
    results = run_world(world, cases, tmp_path, exchange)
    crossed = 0
    for case, res in zip(cases, results):
        o = PyOracle(case["schema"])
        for t in case["tuples"]:
            o.touch(*t)
        want = [P2C[o.check(*q)] for q in case["queries"]]
        got = list(zip(res["perm"], res["err"]))
        assert got == [tuple(w) for w in want], [(q, g, w) for q, g, w in zip(case["queries"], got, want) if tuple(g) != tuple(w)][:5]
        for (rt, pm, st, sid, sr), ids in zip(case["lookups"], res["lookups"]):
            assert set(ids) == o.lookup_resources(rt, pm, st, sid, sr), (rt, pm, st, sid, sr)
        crossed += res["exchanges"]
        if case.get("export_entries") == 1:
            assert res["cap"] > 8, "the export buffer never grew: the redo path was not exercised"
    assert len(set(results[0]["owners"].values())) > 1, "all types landed on one shard: the test exercises nothing"
    assert crossed > 0, "no frontier entry ever crossed a shard boundary"


def test_depth_cases_are_what_they_claim():
    o = PyOracle(CHAIN_SCHEMA)
    for t in chain_case(49):
        o.touch(*t)
    assert o.check("team", "t0", "member", "user", "deep") == "HAS"
    o = PyOracle(CHAIN_SCHEMA)
    for t in chain_case(50):
        o.touch(*t)
    assert o.check("team", "t0", "member", "user", "deep") == "ERR"
