"""Synthetic relationship graphs + request streams of BASELINE.json's configs.

Pure numpy, deterministic per seed.  Used by bench.py and the parity tests; the same
Workload loads into the engine and into the CPU oracle (both expose
add_edges(rtype, rel, stype, srel, res, subj) with dense numeric ids).

C1  flat `namespace#view = viewer + creator`              (plumbing)
C2  cluster -> namespace -> pod arrows, 1 M relationships  (64 k-batch Check)
C3  C2 + 64 power users                                    (Filter / LookupResources)
C4  5-level nested groups, 10 M relationships / 1 M objects (256 k-batch Check, headline)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

SCHEMA_C1 = """
definition user {}
definition namespace {
  relation viewer: user
  relation creator: user
  permission view = viewer + creator
}
"""

SCHEMA_C2 = """
definition user {}
definition cluster {
  relation viewer: user
  relation admin: user
  permission view = viewer + admin
}
definition namespace {
  relation cluster: cluster
  relation viewer: user
  relation creator: user
  permission view = viewer + creator + cluster->view
}
definition pod {
  relation namespace: namespace
  relation viewer: user
  relation creator: user
  permission view = viewer + creator + namespace->view
}
"""

SCHEMA_C4 = """
definition user {}
definition group {
  relation member: user | group#member
}
definition namespace {
  relation viewer: user | group#member
  relation creator: user
  permission view = viewer + creator
}
definition pod {
  relation namespace: namespace
  relation viewer: user | group#member
  relation creator: user
  permission view = viewer + creator + namespace->view
}
"""


@dataclass
class Workload:
    name: str
    schema: str
    edges: list = field(default_factory=list)   # (rtype, rel, stype, srel, res u32[], subj u32[])
    nobjects: dict = field(default_factory=dict)
    # check request stream: all of one (resource type, permission, subject type)
    check: tuple = ("", "", "")                 # (rtype, perm, stype)
    res: np.ndarray = None
    subj: np.ndarray = None
    lookup_subjects: np.ndarray = None          # C3: subject ids for LookupResources(pod, view, user)
    meta: dict = field(default_factory=dict)

    @property
    def ntuples(self) -> int:
        return int(sum(e[4].size for e in self.edges))

    def load(self, target):
        """target: aclgpu.Engine or oracle.orc.Oracle (already constructed with self.schema)."""
        for rt, rel, st, srel, r, s in self.edges:
            target.add_edges(rt, rel, st, srel, r, s)


def _dedup(res, subj):
    key = np.unique(res.astype(np.uint64) << np.uint64(32) | subj.astype(np.uint64))
    return (key >> np.uint64(32)).astype(np.uint32), (key & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def c1(seed: int = 0x5ACE0001, n_ns: int = 1000, n_users: int = 1000, n_checks: int = 100) -> Workload:
    rng = np.random.Generator(np.random.PCG64(seed))
    viewers = np.stack([rng.choice(n_users, size=8, replace=False) for _ in range(n_ns)])
    creators = np.stack([rng.choice(n_users, size=2, replace=False) for _ in range(n_ns)])
    ns8 = np.repeat(np.arange(n_ns, dtype=np.uint32), 8)
    ns2 = np.repeat(np.arange(n_ns, dtype=np.uint32), 2)
    w = Workload("C1", SCHEMA_C1)
    w.edges = [("namespace", "viewer", "user", "", ns8, viewers.reshape(-1).astype(np.uint32)),
               ("namespace", "creator", "user", "", ns2, creators.reshape(-1).astype(np.uint32))]
    w.nobjects = {"namespace": n_ns, "user": n_users}
    half = n_checks // 2
    pick = rng.integers(0, ns8.size, size=half)
    res = np.concatenate([ns8[pick], rng.integers(0, n_ns, size=n_checks - half).astype(np.uint32)])
    subj = np.concatenate([viewers.reshape(-1)[pick].astype(np.uint32), rng.integers(0, n_users, size=n_checks - half).astype(np.uint32)])
    w.check, w.res, w.subj = ("namespace", "view", "user"), res.astype(np.uint32), subj.astype(np.uint32)
    return w


def c2(seed: int = 0x5ACE0002, scale: float = 1.0, batch: int = 65536, power_users: int = 0) -> Workload:
    rng = np.random.Generator(np.random.PCG64(seed))
    n_cl = max(2, int(round(10 * min(1.0, scale * 10))))
    n_ns = max(4, int(1000 * scale))
    n_pod = max(16, int(98990 * scale))
    n_user = max(32, int(10000 * scale))
    pod_ns = rng.integers(0, n_ns, size=n_pod).astype(np.uint32)
    ns_cl = rng.integers(0, n_cl, size=n_ns).astype(np.uint32)
    pods = np.arange(n_pod, dtype=np.uint32)
    nss = np.arange(n_ns, dtype=np.uint32)
    deg = rng.poisson(8.0, size=n_pod)
    pv_r = np.repeat(pods, deg)
    pv_s = rng.integers(0, n_user, size=pv_r.size).astype(np.uint32)
    pv_r, pv_s = _dedup(pv_r, pv_s)
    pod_creator = rng.integers(0, n_user, size=n_pod).astype(np.uint32)
    nv_r = np.repeat(nss, 10)
    nv_s = rng.integers(0, n_user, size=nv_r.size).astype(np.uint32)
    nv_r, nv_s = _dedup(nv_r, nv_s)
    ns_creator = rng.integers(0, n_user, size=n_ns).astype(np.uint32)
    cv_r = np.repeat(np.arange(n_cl, dtype=np.uint32), 20)
    cv_s = rng.integers(0, n_user, size=cv_r.size).astype(np.uint32)
    cv_r, cv_s = _dedup(cv_r, cv_s)
    ca_r = np.repeat(np.arange(n_cl, dtype=np.uint32), 2)
    ca_s = rng.integers(0, n_user, size=ca_r.size).astype(np.uint32)
    ca_r, ca_s = _dedup(ca_r, ca_s)
    w = Workload("C2" if not power_users else "C3", SCHEMA_C2)
    n_user_total = n_user + power_users
    extra_nv = extra_pv = None
    if power_users:  # C3: users seeing ~10 k pods each (100 namespaces + 100 direct grants at scale 1)
        pu = np.arange(n_user, n_user_total, dtype=np.uint32)
        k_ns = max(1, n_ns // 10)
        k_pod = max(1, min(n_pod, 100))
        e_ns = np.concatenate([rng.choice(n_ns, size=k_ns, replace=False) for _ in pu]).astype(np.uint32)
        e_pod = np.concatenate([rng.choice(n_pod, size=k_pod, replace=False) for _ in pu]).astype(np.uint32)
        extra_nv = (e_ns, np.repeat(pu, k_ns))
        extra_pv = (e_pod, np.repeat(pu, k_pod))
        nv_r, nv_s = _dedup(np.concatenate([nv_r, extra_nv[0]]), np.concatenate([nv_s, extra_nv[1]]))
        pv_r, pv_s = _dedup(np.concatenate([pv_r, extra_pv[0]]), np.concatenate([pv_s, extra_pv[1]]))
        w.lookup_subjects = pu
    w.edges = [("pod", "namespace", "namespace", "", pods, pod_ns), ("pod", "viewer", "user", "", pv_r, pv_s),
               ("pod", "creator", "user", "", pods, pod_creator), ("namespace", "cluster", "cluster", "", nss, ns_cl),
               ("namespace", "viewer", "user", "", nv_r, nv_s), ("namespace", "creator", "user", "", nss, ns_creator),
               ("cluster", "viewer", "user", "", cv_r, cv_s), ("cluster", "admin", "user", "", ca_r, ca_s)]
    w.nobjects = {"pod": n_pod, "namespace": n_ns, "cluster": n_cl, "user": n_user_total}
    # request mix: 25 % direct viewer, 25 % namespace-level, 10 % cluster-level, 40 % uniform random
    n1, n2, n3 = batch // 4, batch // 4, batch // 10
    n4 = batch - n1 - n2 - n3
    i1 = rng.integers(0, pv_r.size, size=n1)
    r1, s1 = pv_r[i1], pv_s[i1]
    r2 = rng.integers(0, n_pod, size=n2).astype(np.uint32)
    # a viewer of the pod's namespace: pick any namespace-viewer edge of that namespace
    order = np.argsort(nv_r, kind="stable")
    nv_r_s, nv_s_s = nv_r[order], nv_s[order]
    start = np.searchsorted(nv_r_s, pod_ns[r2], side="left")
    end = np.searchsorted(nv_r_s, pod_ns[r2], side="right")
    has = end > start
    pick = start + (rng.integers(0, 1 << 30, size=n2) % np.maximum(end - start, 1))
    s2 = np.where(has, nv_s_s[np.minimum(pick, nv_s_s.size - 1)], ns_creator[pod_ns[r2]]).astype(np.uint32)
    r3 = rng.integers(0, n_pod, size=n3).astype(np.uint32)
    ocl = np.argsort(cv_r, kind="stable")
    cv_r_s, cv_s_s = cv_r[ocl], cv_s[ocl]
    cl = ns_cl[pod_ns[r3]]
    st3 = np.searchsorted(cv_r_s, cl, side="left")
    en3 = np.searchsorted(cv_r_s, cl, side="right")
    pk3 = st3 + (rng.integers(0, 1 << 30, size=n3) % np.maximum(en3 - st3, 1))
    s3 = cv_s_s[np.minimum(pk3, cv_s_s.size - 1)].astype(np.uint32)
    r4 = rng.integers(0, n_pod, size=n4).astype(np.uint32)
    s4 = rng.integers(0, n_user, size=n4).astype(np.uint32)
    res = np.concatenate([r1, r2, r3, r4]).astype(np.uint32)
    subj = np.concatenate([s1, s2, s3, s4]).astype(np.uint32)
    perm = rng.permutation(batch)
    w.check, w.res, w.subj = ("pod", "view", "user"), res[perm], subj[perm]
    return w


def c3(seed: int = 0x5ACE0003, scale: float = 1.0, batch: int = 4096, power_users: int = 64) -> Workload:
    return c2(seed, scale, batch, power_users)


def c4(seed: int = 0x5ACE0004, scale: float = 1.0, batch: int = 262144, n_user: int = 0) -> Workload:
    rng = np.random.Generator(np.random.PCG64(seed))
    n_user = n_user or max(64, int(100_000 * scale))
    levels = 5
    per_level = max(4, int(10_000 * scale))
    n_group = per_level * levels
    n_ns = max(4, int(5_000 * scale))
    n_pod = max(32, int(845_000 * scale))
    target = int(10_000_000 * scale)
    # nested groups: a level-l group contains 1-4 groups of level l+1 (acyclic by construction)
    parents = np.arange(per_level * (levels - 1), dtype=np.uint32)
    fan = rng.integers(1, 5, size=parents.size)
    gg_r = np.repeat(parents, fan)
    gg_s = ((gg_r // per_level + 1) * per_level + rng.integers(0, per_level, size=gg_r.size)).astype(np.uint32)
    gg_r, gg_s = _dedup(gg_r, gg_s)
    # user membership: ~58 users per group
    n_gu = max(n_group, int(3_000_000 * scale) - gg_r.size)
    gu_r = rng.integers(0, n_group, size=n_gu).astype(np.uint32)
    gu_s = rng.integers(0, n_user, size=n_gu).astype(np.uint32)
    gu_r, gu_s = _dedup(gu_r, gu_s)
    pods = np.arange(n_pod, dtype=np.uint32)
    nss = np.arange(n_ns, dtype=np.uint32)
    pod_ns = rng.integers(0, n_ns, size=n_pod).astype(np.uint32)
    pod_creator = rng.integers(0, n_user, size=n_pod).astype(np.uint32)
    ns_creator = rng.integers(0, n_user, size=n_ns).astype(np.uint32)
    remaining = max(4 * (n_pod + n_ns), target - gg_r.size - gu_r.size - 2 * n_pod - n_ns)
    # viewers: split 50/50 between users and top-level (level 0) groups, spread over pods and namespaces
    n_obj = n_pod + n_ns
    half = remaining // 2
    vu_o = rng.integers(0, n_obj, size=half)
    vu_s = rng.integers(0, n_user, size=half).astype(np.uint32)
    vg_o = rng.integers(0, n_obj, size=remaining - half)
    vg_s = rng.integers(0, per_level, size=remaining - half).astype(np.uint32)  # level-0 groups: every group check walks 5 levels

    def split(o, s):
        is_pod = o < n_pod
        return _dedup(o[is_pod].astype(np.uint32), s[is_pod]), _dedup((o[~is_pod] - n_pod).astype(np.uint32), s[~is_pod])

    (pvu_r, pvu_s), (nvu_r, nvu_s) = split(vu_o, vu_s)
    (pvg_r, pvg_s), (nvg_r, nvg_s) = split(vg_o, vg_s)
    w = Workload("C4", SCHEMA_C4)
    w.edges = [("group", "member", "group", "member", gg_r, gg_s), ("group", "member", "user", "", gu_r, gu_s),
               ("pod", "namespace", "namespace", "", pods, pod_ns), ("pod", "creator", "user", "", pods, pod_creator),
               ("namespace", "creator", "user", "", nss, ns_creator), ("pod", "viewer", "user", "", pvu_r, pvu_s),
               ("pod", "viewer", "group", "member", pvg_r, pvg_s), ("namespace", "viewer", "user", "", nvu_r, nvu_s),
               ("namespace", "viewer", "group", "member", nvg_r, nvg_s)]
    w.nobjects = {"user": n_user, "group": n_group, "namespace": n_ns, "pod": n_pod}
    # request mix: hit depth uniform over {1..5 levels of nesting, via-namespace, uniform random}
    cats = rng.integers(0, 7, size=batch)
    res = rng.integers(0, n_pod, size=batch).astype(np.uint32)
    subj = rng.integers(0, n_user, size=batch).astype(np.uint32)

    def pick_from(sorted_r, sorted_s, keys, r):
        st = np.searchsorted(sorted_r, keys, side="left")
        en = np.searchsorted(sorted_r, keys, side="right")
        ok = en > st
        p = st + (r % np.maximum(en - st, 1))
        return ok, sorted_s[np.minimum(p, sorted_s.size - 1)]

    # depth-k hits: start from a pod->group viewer edge, descend k-1 nested levels, pick a user member
    for k in range(1, levels + 1):
        idx = np.flatnonzero(cats == k - 1)
        if not idx.size or not pvg_r.size:
            continue
        e = rng.integers(0, pvg_r.size, size=idx.size)
        pod, g = pvg_r[e], pvg_s[e].copy()
        alive = np.ones(idx.size, dtype=bool)
        for _ in range(k - 1):
            ok, child = pick_from(gg_r, gg_s, g, rng.integers(0, 1 << 30, size=idx.size))
            alive &= ok
            g = np.where(ok, child, g)
        ok, u = pick_from(gu_r, gu_s, g, rng.integers(0, 1 << 30, size=idx.size))
        alive &= ok
        res[idx] = np.where(alive, pod, res[idx])
        subj[idx] = np.where(alive, u, subj[idx])
    idx = np.flatnonzero(cats == 5)  # via the pod's namespace: a direct user viewer of it
    if idx.size and nvu_r.size:
        ok, u = pick_from(nvu_r, nvu_s, pod_ns[res[idx]], rng.integers(0, 1 << 30, size=idx.size))
        subj[idx] = np.where(ok, u, subj[idx])
    w.check, w.res, w.subj = ("pod", "view", "user"), res, subj
    w.meta = {"levels": levels, "groups_per_level": per_level}
    return w


def c5(seed: int = 0x5ACE0005, scale: float = 1.0, batch: int = 262144, n_lookups: int = 64, n_user: int = 0) -> Workload:
    """BASELINE config 5: the C4 generator scaled x10 (100 M relationships / 10 M objects at scale 1.0), for the graph
    sharded by object type; `stream(k)` yields the mixed request stream: 90 % Check batches, 10 % Filter requests."""
    w = c4(seed=seed, scale=10.0 * scale, batch=batch, n_user=n_user)
    w.name = "C5"
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xF17E))
    w.lookup_subjects = rng.integers(0, w.nobjects["user"], size=n_lookups).astype(np.uint32)
    w.meta["stream_seed"] = seed
    return w


def c5_stream(w: Workload, steps: int):
    """['C' | ('F', subject id)] * steps, interleaved by the workload's seed; at least one of each kind when steps >= 2."""
    rng = np.random.Generator(np.random.PCG64(w.meta.get("stream_seed", 0x5ACE0005)))
    ops = [("F", int(w.lookup_subjects[i % w.lookup_subjects.size])) if rng.random() < 0.1 else "C" for i in range(steps)]
    if steps >= 2:
        if all(o == "C" for o in ops):
            ops[steps // 2] = ("F", int(w.lookup_subjects[0]))
        if all(o != "C" for o in ops):
            ops[0] = "C"
    return ops


def by_name(name: str, **kw) -> Workload:
    return {"C1": c1, "C2": c2, "C3": c3, "C4": c4, "C5": c5}[name.upper()](**kw)
