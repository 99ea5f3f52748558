#!/usr/bin/env python3
"""usage (GPU box): tools/fuzz_gpu.py [--seed S] [--steps N] -- differential fuzz of the engine against the CPU oracle on a LIVE graph.

A nested-group / arrow graph (the C4 schema, cycles allowed) is mutated step by step -- TOUCH / CREATE / DELETE batches, filter deletes --
and between the writes both sides answer the same random Check batches (1 ... 70 000 items: the single-launch walk's 4-wave and 16-wave
kernels, the host-mapped small-batch path, the micro-batcher's single checks) and LookupResources requests (single and batched).  Every
answer, every error code and every allowed-id set must be equal.  Writes land in the device snapshot as in-place patches, background
compactions happen when the headroom runs low: the fuzz is what exercises patch -> query -> patch sequences nobody wrote a test for.
The oracle is the checker (tests / tools only).  Exit code 1 and a dump of the first difference on a mismatch."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]


# The C4 schema with every non-monotone construct the engine compiles into combine programs (plan.hpp BX_*): exclusion at the top of a permission
# that arrows reach, intersection over an arrow, a userset subject that is itself a non-monotone permission (group#active), wildcards on both
# sides of a `-`, and a permission that subtracts an intersection.  Cycles through group#member (and through group#active, via pod viewers) stay
# legal: a non-monotone cycle ends at the depth limit on both sides.  `everywhere` / `vetted`: intersection arrows over a pod's namespaces (a
# fifth of the pod#namespace relationships name a second namespace).
SCHEMA_COMBINE = """
definition user {}
definition group {
  relation member: user | group#member
  relation banned: user
  permission active = member - banned
}
definition namespace {
  relation viewer: user | group#member | user:*
  relation creator: user
  relation banned: user | group#member
  permission view = (viewer + creator) - banned
}
definition pod {
  relation namespace: namespace
  relation viewer: user | group#member | group#active
  relation creator: user
  relation auditor: user | group#member
  relation banned: user | user:*
  permission view = (viewer + creator + namespace->view) - banned
  permission audit = auditor & namespace->view
  permission edit = creator + (viewer & auditor) - banned
  permission hidden = view - (auditor & creator)
  permission everywhere = namespace.all(view)
  permission vetted = creator + namespace.all(view) & viewer
}
"""


def run(seed: int, steps: int, big: int = 70000, verbose: bool = True, burst: int = 25, universe: int = 1, compact_early: bool = False, schema: str = "c4",
        recycle: bool = False, acyclic: bool = False) -> dict:
    import aclgpu
    from aclgpu import workloads
    from oracle import orc
    combine = schema == "combine"
    schema_text = SCHEMA_COMBINE if combine else workloads.SCHEMA_C4

    rng = random.Random(seed)
    users = [f"u{i}" for i in range(160 * universe)]
    groups = [f"g{i}" for i in range(48 * universe)]
    nss = [f"n{i}" for i in range(8 * universe)]
    pods = [f"{rng.choice(nss)}/p{i}" for i in range(240 * universe)]

    fresh = [0]

    def rand_tuple():
        if recycle and rng.random() < 0.15:  # objects nobody has named before: they take over the ids of objects that lost their last relationship
            fresh[0] += 1
            return rng.choice([f"pod:{rng.choice(nss)}/fresh{fresh[0]}#viewer@user:{rng.choice(users)}", f"pod:{rng.choice(pods)}#viewer@user:newcomer{fresh[0]}",
                               f"group:team{fresh[0]}#member@user:{rng.choice(users)}", f"pod:{rng.choice(pods)}#viewer@group:team{fresh[0] - rng.randrange(3)}#member"])
        if combine and rng.random() < 0.3:  # the relations only the combine schema has
            k = rng.randrange(9)
            if k == 0: return f"group:{rng.choice(groups)}#banned@user:{rng.choice(users)}"
            if k == 1: return f"namespace:{rng.choice(nss)}#banned@user:{rng.choice(users)}"
            if k == 2: return f"namespace:{rng.choice(nss)}#banned@group:{rng.choice(groups)}#member"
            if k == 3: return f"namespace:{rng.choice(nss)}#viewer@user:*" if rng.random() < 0.3 else f"pod:{rng.choice(pods)}#viewer@group:{rng.choice(groups)}#active"
            if k == 4: return f"pod:{rng.choice(pods)}#auditor@user:{rng.choice(users)}"
            if k == 5: return f"pod:{rng.choice(pods)}#auditor@group:{rng.choice(groups)}#member"
            if k == 6: return f"pod:{rng.choice(pods)}#banned@user:{rng.choice(users)}"
            if k == 7: return f"pod:{rng.choice(pods)}#banned@user:*" if rng.random() < 0.2 else f"pod:{rng.choice(pods)}#banned@user:{rng.choice(users)}"
            return f"pod:{rng.choice(pods)}#viewer@group:{rng.choice(groups)}#active"
        k = rng.randrange(9)
        if k == 0 and acyclic:  # (--acyclic: a group only holds groups behind it in the list -- no Check ends at the depth limit, one-user calls take the reverse walk)
            i, j = sorted(rng.sample(range(len(groups)), 2))
            return f"group:{groups[i]}#member@group:{groups[j]}#member"
        if k == 0: return f"group:{rng.choice(groups)}#member@group:{rng.choice(groups)}#member"   # (cycles welcome)
        if k == 1: return f"group:{rng.choice(groups)}#member@user:{rng.choice(users)}"
        if k == 2: return f"pod:{(p := rng.choice(pods))}#namespace@namespace:{p.split('/')[0] if rng.random() < 0.8 else rng.choice(nss)}"
        if k == 3: return f"pod:{rng.choice(pods)}#creator@user:{rng.choice(users)}"
        if k == 4: return f"namespace:{rng.choice(nss)}#creator@user:{rng.choice(users)}"
        if k == 5: return f"pod:{rng.choice(pods)}#viewer@user:{rng.choice(users)}"
        if k == 6: return f"pod:{rng.choice(pods)}#viewer@group:{rng.choice(groups)}#member"
        if k == 7: return f"namespace:{rng.choice(nss)}#viewer@user:{rng.choice(users)}"
        return f"namespace:{rng.choice(nss)}#viewer@group:{rng.choice(groups)}#member"

    def rand_query():
        k = rng.randrange(10)
        u = rng.choice(users) if rng.random() < 0.97 else "stranger"
        if combine and rng.random() < 0.45:
            kk = rng.randrange(6)
            if kk < 3: return ("pod", rng.choice(pods), rng.choice(("audit", "edit", "hidden", "everywhere", "vetted")), "user", u, "")
            if kk == 3: return ("group", rng.choice(groups), "active", "user", u, "")
            if kk == 4: return ("pod", rng.choice(pods), "view", "group", rng.choice(groups), "active")  # a non-monotone userset as the subject
            return ("namespace", rng.choice(nss), "view", "group", rng.choice(groups), "member")
        if k < 6: return ("pod", rng.choice(pods) if rng.random() < 0.97 else "n0/ghost", "view", "user", u, "")
        if k < 8: return ("namespace", rng.choice(nss), "view", "user", u, "")
        if k < 9: return ("group", rng.choice(groups), "member", "user", u, "")
        return ("pod", rng.choice(pods), "view", "group", rng.choice(groups), "member")  # a userset as the subject

    if recycle:
        os.environ["ACL_ID_QUARANTINE_MS"] = "0"  # (read when the schema is loaded) freed ids are reused by the next new name at once
    if compact_early:
        os.environ["ACL_COMPACTION_SLACK"] = "0"  # (read at acl_open) background compactions, adopted with the writes since replayed, on this small graph too
    try:
        e = aclgpu.Engine(schema_text)
    finally:
        os.environ.pop("ACL_COMPACTION_SLACK", None)
        os.environ.pop("ACL_ID_QUARANTINE_MS", None)
    o = orc.Oracle(schema_text)
    live = set()
    init = list(dict.fromkeys(rand_tuple() for _ in range(2500 * universe)))
    for i in range(0, len(init), 500):
        o.write([(orc.OP_TOUCH, t) for t in init[i:i + 500]])
        e.write([(aclgpu.OP_TOUCH, t) for t in init[i:i + 500]])
    live.update(init)
    stats = {"writes": 0, "write_errors": 0, "filter_deletes": 0, "checks": 0, "lookups": 0, "single_checks": 0}
    e.batcher_start(4096, 100)
    t0 = time.time()
    for step in range(steps):
        r = rng.random()
        if r < 0.45:  # ---- a write batch: both sides accept it or reject it with the same code
            ups = []
            pool = sorted(live) if live else []  # (once per batch: a delete picks from the relationships that exist)
            for _ in range(rng.randrange(1, burst)):
                q = rng.random()
                if q < 0.5: ups.append((aclgpu.OP_TOUCH, rand_tuple()))
                elif q < 0.5 + 0.1 / max(1, burst // 25): ups.append((aclgpu.OP_CREATE, rand_tuple()))
                elif pool: ups.append((aclgpu.OP_DELETE, rng.choice(pool) if rng.random() < 0.9 else rand_tuple()))
            ups = list({t: (op, t) for op, t in ups}.values())  # one update per relationship in a request
            eo = ee = None
            try:
                o.write([({aclgpu.OP_TOUCH: orc.OP_TOUCH, aclgpu.OP_CREATE: orc.OP_CREATE, aclgpu.OP_DELETE: orc.OP_DELETE}[op], t) for op, t in ups])
            except orc.OracleError as x:
                eo = x.code
            try:
                e.write(ups)
            except aclgpu.AclError as x:
                ee = x.code
            assert eo == ee, f"step {step}: write outcome differs: oracle {eo}, engine {ee}: {ups}"
            stats["writes"] += 1
            if eo is None:
                for op, t in ups:
                    (live.discard if op == aclgpu.OP_DELETE else live.add)(t)
            else:
                stats["write_errors"] += 1
        elif r < 0.5:  # ---- DeleteRelationships by filter
            f = rng.choice([dict(rtype="pod", rid=rng.choice(pods)), dict(rtype="group", rel="member", stype="user", sid=rng.choice(users)),
                            dict(rtype="namespace", rid=rng.choice(nss), rel="viewer")] +
                           ([dict(rtype="pod", rel="banned", stype="user", sid="*"), dict(rtype="namespace", rel="banned")] if combine else []))
            n1, n2 = o.delete_by_filter(**f), e.delete_by_filter(**f)
            assert n1 == n2, f"step {step}: delete_by_filter {f}: oracle removed {n1}, engine {n2}"
            live = set(f"{a}:{b}#{c}@{d}:{x}" + (f"#{y}" if y else "") for t in ("group", "namespace", "pod") for a, b, c, d, x, y, _ in o.read(rtype=t))
            stats["filter_deletes"] += 1
        elif r < 0.56 and not combine:  # ---- PostFilter's shape: ONE user's pairs -- CheckBulkPermissions itself twice (the second call may sweep the type for depth
            # errors and take the reverse walk: engine.cpp no_object_is_deep; with a cycle behind some resource the calls stay forward and carry its depth errors),
            # then the keep mask of the same pairs.  Every permissionship, every error and every keep byte against the oracle.
            rt, perm, ids = rng.choice([("pod", "view", pods), ("namespace", "view", nss), ("group", "member", groups)])
            u = rng.choice(users) if rng.random() < 0.9 else "stranger"
            k_ = rng.choice([520, 900, 2500])
            qs = [(rt, rng.choice(ids) if rng.random() < 0.98 else "n0/ghost", perm, "user", u, "") for _ in range(k_)]
            want = {q: o.check(*q) for q in set(qs)}
            for rep in range(2):
                perms, errs = e.check_bulk(qs)
                for i, q in enumerate(qs):
                    assert (perms[i], errs[i]) == tuple(want[q]), f"step {step}: one-user check {q} (item {i} of {k_}, call {rep}): engine {(perms[i], errs[i])}, oracle {want[q]}"
            keep = e.check_bulk_keep(qs, list(range(k_ + 1)))
            for i, q in enumerate(qs):
                assert bool(keep[i]) == (tuple(want[q]) == (2, 0)), f"step {step}: keep mask of {q} (item {i}): engine {keep[i]}, oracle {want[q]}"
            stats["checks"] += 3 * k_
            stats["one_user_calls"] = stats.get("one_user_calls", 0) + 3
        elif r < 0.85:  # ---- a Check batch
            n = rng.choice([1, 1, 7, 64, 64, 900, 5000, big if step % 7 == 3 else 3000])
            qs = [rand_query() for _ in range(min(n, 6000))]
            qs = (qs * (n // len(qs) + 1))[:n]
            want = {q: o.check(*q) for q in set(qs)}
            if n == 1 and rng.random() < 0.5:  # the micro-batcher's single check
                got = [e.check_one(*qs[0])]
                stats["single_checks"] += 1
            else:
                perms, errs = e.check_bulk(qs)
                got = list(zip(perms, errs))
            for i, q in enumerate(qs):
                assert tuple(got[i]) == tuple(want[q]), f"step {step}: check {q} (item {i} of {n}): engine {got[i]}, oracle {want[q]}"
            stats["checks"] += n
        else:  # ---- LookupResources: one subject, or a batch of subjects in one walk
            rt, perm = rng.choice([("pod", "view"), ("namespace", "view"), ("group", "member")] +
                                  ([("pod", "audit"), ("pod", "edit"), ("pod", "hidden"), ("group", "active"), ("pod", "everywhere"), ("pod", "vetted")] if combine else []))
            subs = rng.sample(users, rng.choice([1, 1, 3, 20]))
            def out(fn, *a):  # ("ok", ids) or ("err", code): a lookup whose candidate's Check errs fails as a whole, in the engine and in the oracle
                try:
                    return ("ok", sorted(fn(*a)))
                except Exception as ex:  # noqa: BLE001
                    if getattr(ex, "code", None) is None:
                        raise
                    return ("err", ex.code)

            for u in subs[:3]:
                a, b = out(e.lookup, rt, perm, "user", u), out(o.lookup, rt, perm, "user", u)
                assert a == b, f"step {step}: lookup {rt}#{perm}@user:{u}: engine {a[0]} {str(a[1])[:200]}, oracle {b[0]} {str(b[1])[:200]}"
                stats["lookups_failed"] = stats.get("lookups_failed", 0) + int(a[0] == "err")
            if len(subs) > 3:
                ids = [e.intern("user", u) for u in subs]
                wants = [out(o.lookup, rt, perm, "user", u) for u in subs]
                batch = out(lambda: [0] if e.lookup_ids_batch(rt, perm, "user", "", ids) is None else [1])
                if any(w_[0] == "err" for w_ in wants):  # one failing lookup fails the batch call
                    assert batch[0] == "err", f"step {step}: batched lookup {rt}#{perm}: the oracle fails {[u for u, w_ in zip(subs, wants) if w_[0] == 'err'][:3]}, the engine answered"
                else:
                    bm, counts = e.lookup_ids_batch(rt, perm, "user", "", ids)
                    for j, u in enumerate(subs):
                        got = sorted(e.object_name(rt, i) for i in range(e.object_count(rt)) if (bm[j][i >> 5] >> (i & 31)) & 1)
                        assert got == wants[j][1], f"step {step}: batched lookup {rt}#{perm}@user:{u}"
            stats["lookups"] += len(subs)
        if verbose and step % 50 == 49:
            print(f"step {step + 1}/{steps}: {stats}, {len(live)} relationships, {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    st = e.stats()
    stats.update({k: int(st[k]) for k in ("snapshot_builds", "snapshot_patches", "snapshot_compactions", "local_passes", "rev_local_passes", "ids_recycled", "keep_route_calls", "depth_sweeps") if k in st})
    e.close()
    return stats


def run_patcher(seed: int, steps: int, universe: int = 1, burst: int = 25, shard=None, schema: str = "c4") -> dict:
    """No GPU: the same kind of write stream on a STORE-ONLY engine, and after every write the host snapshot is brought up to date the way a read
    would (patched in place, or rebuilt) and verified against the store (acl_selfcheck_snapshot: every relationship findable by the kernels'
    search, nothing dead left, rows sorted, no unsound leaf flag).  -> how often it was patched (1), rebuilt (0) or current (2)."""
    import aclgpu
    from aclgpu import workloads

    rng = random.Random(seed)
    users = [f"u{i}" for i in range(160 * universe)]
    groups = [f"g{i}" for i in range(48 * universe)]
    nss = [f"n{i}" for i in range(8 * universe)]
    pods = [f"{rng.choice(nss)}/p{i}" for i in range(240 * universe)]
    shapes = [lambda: f"group:{rng.choice(groups)}#member@group:{rng.choice(groups)}#member", lambda: f"group:{rng.choice(groups)}#member@user:{rng.choice(users)}",
              lambda: f"pod:{rng.choice(pods)}#namespace@namespace:{rng.choice(nss)}", lambda: f"pod:{rng.choice(pods)}#creator@user:{rng.choice(users)}",
              lambda: f"namespace:{rng.choice(nss)}#creator@user:{rng.choice(users)}", lambda: f"pod:{rng.choice(pods)}#viewer@user:{rng.choice(users)}",
              lambda: f"pod:{rng.choice(pods)}#viewer@group:{rng.choice(groups)}#member", lambda: f"namespace:{rng.choice(nss)}#viewer@user:{rng.choice(users)}",
              lambda: f"namespace:{rng.choice(nss)}#viewer@group:{rng.choice(groups)}#member"]
    if schema == "combine":  # wildcard classes, a non-monotone userset subject, the relations behind `-` and `&`
        shapes += [lambda: f"group:{rng.choice(groups)}#banned@user:{rng.choice(users)}", lambda: f"namespace:{rng.choice(nss)}#banned@group:{rng.choice(groups)}#member",
                   lambda: f"namespace:{rng.choice(nss)}#viewer@user:*", lambda: f"pod:{rng.choice(pods)}#viewer@group:{rng.choice(groups)}#active",
                   lambda: f"pod:{rng.choice(pods)}#auditor@group:{rng.choice(groups)}#member", lambda: f"pod:{rng.choice(pods)}#banned@user:*",
                   lambda: f"pod:{rng.choice(pods)}#banned@user:{rng.choice(users)}", lambda: f"pod:{rng.choice(pods)}#auditor@user:{rng.choice(users)}"]
    e = aclgpu.Engine(SCHEMA_COMBINE if schema == "combine" else workloads.SCHEMA_C4, store_only=True)
    if shard:  # (rank, world): the snapshot of ONE shard of the type-hash layout -- it holds, and patches, only the rows of the types it owns
        e._check(e._L.acl_shard_configure(e._h, shard[0], shard[1]))
    live = set(dict.fromkeys(rng.choice(shapes)() for _ in range(2500 * universe)))
    init = sorted(live)
    for i in range(0, len(init), 500):
        e.write([(aclgpu.OP_TOUCH, t) for t in init[i:i + 500]])
    e.selfcheck_snapshot_code()
    codes = {0: 0, 1: 0, 2: 0, "adopted": 0, "dropped": 0}
    pending_build = None
    for _ in range(steps):
        if rng.random() < 0.9:
            pool = sorted(live)
            ups = {}
            for _u in range(rng.randrange(1, burst)):
                t = rng.choice(shapes)() if rng.random() < 0.55 or not pool else rng.choice(pool)
                ups[t] = (aclgpu.OP_TOUCH if t not in live or rng.random() < 0.3 else aclgpu.OP_DELETE, t)
            e.write(list(ups.values()))
            for op, t in ups.values():
                (live.discard if op == aclgpu.OP_DELETE else live.add)(t)
        else:
            e.delete_by_filter(**rng.choice([dict(rtype="pod", rid=rng.choice(pods)), dict(rtype="group", rel="member", stype="user", sid=rng.choice(users))]))
            live = set(f"{a}:{b}#{c}@{d}:{x}" + (f"#{y}" if y else "") for t in ("group", "namespace", "pod") for a, b, c, d, x, y, *_ in e.read(rtype=t))
        codes[e.selfcheck_snapshot_code()] += 1  # (raises when the snapshot does not describe the store)
        # a background compaction's two halves around the writes in between: build from a copy-on-write view now, adopt it some writes later
        if pending_build is None and rng.random() < 0.02:
            e.selfcheck_compaction(0)
            pending_build = rng.randrange(0, 30)
        elif pending_build is not None:
            if pending_build == 0:
                codes["adopted" if e.selfcheck_compaction(1) else "dropped"] += 1  # (phase 1 verifies the adopted snapshot against the store)
                pending_build = None
            else:
                pending_build -= 1
    e.close()
    return codes


def run_expiry(seed: int, steps: int, verbose: bool = True) -> dict:
    """The reference's own schema (pkg/spicedb/bootstrap.yaml) under the dual write's shapes: lock tuples created behind MUST_NOT_MATCH and deleted
    again (workflow.go:392-462), idempotency keys that EXPIRE (activity.go:81-102), payload relationships -- while the clock moves forwards in
    random steps (seconds to days: keys run out in between, expired ones are collected after 24 h).  After every step a batch of checks on live,
    expired and never-written keys, locks and payloads, and ReadRelationships of the expiring class, must equal the oracle's."""
    import json

    import aclgpu
    from oracle import orc

    rng = random.Random(seed)
    b = json.load(open(os.path.join(ROOT, "tests", "golden", "bootstrap.json")))
    e = aclgpu.Engine(b["schema"], "\n".join(b["relationships"]))
    o = orc.Oracle(b["schema"])
    o.write([(orc.OP_TOUCH, r) for r in b["relationships"]])
    now = 1_700_000_000
    e.set_now(now)
    o.set_now(now)
    wfs = [f"wf{i}" for i in range(40)]
    acts = [f"act{i:x}" for i in range(120)]
    locks = [f"lk{i:x}" for i in range(30)]
    users = [f"u{i}" for i in range(20)]
    nss = [f"ns{i}" for i in range(12)]
    stats = {"writes": 0, "write_errors": 0, "clock_moves": 0, "checks": 0, "reads": 0}
    OPS = {aclgpu.OP_TOUCH: orc.OP_TOUCH, aclgpu.OP_CREATE: orc.OP_CREATE, aclgpu.OP_DELETE: orc.OP_DELETE}
    for step in range(steps):
        r = rng.random()
        if r < 0.5:
            ups, pre = [], []
            k = rng.random()
            if k < 0.45:    # the pessimistic dual write's first request: payload + lock CREATE behind MUST_NOT_MATCH + an expiring idempotency key
                lk, wf = rng.choice(locks), rng.choice(wfs)
                pre = [(aclgpu.PRE_MUST_NOT_MATCH, dict(rtype="lock", rid=lk, rel="workflow"))]
                ups = [(aclgpu.OP_TOUCH, f"namespace:{rng.choice(nss)}#creator@user:{rng.choice(users)}"),
                       (aclgpu.OP_CREATE, f"lock:{lk}#workflow@workflow:{wf}"),
                       (aclgpu.OP_TOUCH, f"workflow:{wf}#idempotency_key@activity:{rng.choice(acts)}", now + rng.choice([1, 5, 60, 3600, 86400, 200000]))]
            elif k < 0.8:   # ... and its second: the lock goes away
                ups = [(aclgpu.OP_DELETE, f"lock:{rng.choice(locks)}#workflow@workflow:{rng.choice(wfs)}")]
            else:           # keys rewritten with a new expiry, or none
                ups = [(aclgpu.OP_TOUCH, f"workflow:{rng.choice(wfs)}#idempotency_key@activity:{rng.choice(acts)}", rng.choice([0, now + 2, now + 30, now - 5]))
                       for _ in range(rng.randrange(1, 6))]
                ups = list({u[1]: u for u in ups}.values())
            eo = ee = None
            try:
                o.write([(OPS[u[0]],) + tuple(u[1:]) for u in ups], [({aclgpu.PRE_MUST_NOT_MATCH: orc.PRE_MUST_NOT_MATCH}[p[0]], p[1]) for p in pre])
            except orc.OracleError as x:
                eo = x.code
            try:
                e.write(ups, pre)
            except aclgpu.AclError as x:
                ee = x.code
            assert eo == ee, f"step {step}: write outcome differs: oracle {eo}, engine {ee}: {ups} {pre}"
            stats["writes"] += 1
            stats["write_errors"] += eo is not None
        elif r < 0.7:
            now += rng.choice([1, 1, 3, 10, 61, 3601, 50000, 90000])
            e.set_now(now)
            o.set_now(now)
            stats["clock_moves"] += 1
        qs = [("workflow", rng.choice(wfs), "idempotency_key", "activity", rng.choice(acts), "") for _ in range(150)] + \
             [("lock", rng.choice(locks), "workflow", "workflow", rng.choice(wfs), "") for _ in range(40)] + \
             [("namespace", rng.choice(nss), rng.choice(["view", "edit", "admin", "no_one_at_all"]), "user", rng.choice(users + ["rakis"]), "") for _ in range(40)] + \
             [("namespace", "spicedb-kubeapi-proxy", "view", "user", "rakis", "")]
        perms, errs = e.check_bulk(qs)
        for i, q in enumerate(qs):
            assert (perms[i], errs[i]) == tuple(o.check(*q)), f"step {step} (now {now}): check {q}: engine {(perms[i], errs[i])}, oracle {o.check(*q)}"
        stats["checks"] += len(qs)
        if step % 5 == 0:  # ReadRelationships of the expiring class (activity.go:107-149 reads the keys back): expired ones are gone on both sides
            a, b2 = sorted(e.read(rtype="workflow")), sorted(o.read(rtype="workflow"))
            assert [x[:6] for x in a] == [x[:6] for x in b2], f"step {step} (now {now}): workflow relationships differ: {set(a) ^ set(b2)}"
            stats["reads"] += 1
        if verbose and step % 100 == 99:
            print(f"step {step + 1}/{steps}: {stats}, now +{now - 1_700_000_000} s", file=sys.stderr, flush=True)
    st = e.stats()
    stats.update({k: int(st[k]) for k in ("snapshot_builds", "snapshot_patches", "snapshot_compactions") if k in st})
    e.close()
    return stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--burst", type=int, default=25, help="updates per write, at most (<= 1000: the reference's limit, spicedb.go:35)")
    ap.add_argument("--compact-early", action="store_true", help="ACL_COMPACTION_SLACK=0: background compactions (and their adoption with the writes since replayed) happen on this small graph too")
    ap.add_argument("--patcher", action="store_true", help="no GPU: a store-only engine whose host snapshot is verified against the store after every write")
    ap.add_argument("--expiry", action="store_true", help="the other campaign: the reference's bootstrap schema, dual-write shapes, expiring idempotency keys, a moving clock")
    ap.add_argument("--schema", choices=["c4", "combine"], default="c4", help="combine: the C4 schema with exclusions, intersections, wildcards and a non-monotone userset subject")
    ap.add_argument("--recycle", action="store_true", help="ACL_ID_QUARANTINE_MS=0 and a stream of never-seen object names in the writes: ids of objects that lost their last relationship are taken over under the reads")
    ap.add_argument("--acyclic", action="store_true", help="group nesting without cycles: one-user CheckBulkPermissions calls take the reverse walk once the depth sweep has run")
    ap.add_argument("--universe", type=int, default=1, help="scale of the object universe (x 160 users, 48 groups, 8 namespaces, 240 pods)")
    a = ap.parse_args()
    try:
        print(run_patcher(a.seed, a.steps, a.universe, a.burst, schema=a.schema) if a.patcher else run_expiry(a.seed, a.steps) if a.expiry
              else run(a.seed, a.steps, burst=a.burst, universe=a.universe, compact_early=a.compact_early, schema=a.schema, recycle=a.recycle, acyclic=a.acyclic))
    except AssertionError as x:
        print("MISMATCH:", x)
        sys.exit(1)
