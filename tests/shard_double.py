"""CpuShard -- TEST DOUBLE of one graph shard (test infrastructure, never shipped or benchmarked).

The sharded protocol (aclgpu/sharded.py: step -> all-gather counts -> all-gather exports -> import ->
reduce) is host logic that must be covered on CPU with world_size-2 gloo.  The HIP engine cannot run here,
so this double plays the shard: a level-synchronous, per-shard evaluator over the Python oracle's
relationship store (oracle/pyoracle.py), speaking the same stepper interface and the same 16-byte entry
format as aclgpu.sharded.GpuShard.  Ownership rule identical to the engine's: fnv1a(type) mod world.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle.pyoracle import MAX_DEPTH, Permission, PyOracle, Relation

ITEM_DTYPE = np.dtype([("resource_type", "<u2"), ("permission", "<u2"), ("resource_id", "<u4"), ("subject_type", "<u2"),
                       ("subject_relation", "<u2"), ("subject_id", "<u4")])
NO_RELATION = 0xFFFF
FOREIGN = 1 << 19


def fnv1a(s: str) -> int:
    h = 2166136261
    for c in s.encode():
        h = ((h ^ c) * 16777619) & 0xFFFFFFFF
    h ^= h >> 16  # avalanche (murmur's 32-bit finaliser), as plan.cpp shard_of_type
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


class Universe:
    """Deterministic ids shared by every rank and by the test: types/members in schema order, objects sorted."""

    def __init__(self, schema: str, tuples):
        self.o = PyOracle(schema)
        for t in tuples:
            self.o.touch(*t)
        self.types = list(self.o.defs)
        self.tid = {t: i for i, t in enumerate(self.types)}
        self.members = {t: list(self.o.defs[t].members) for t in self.types}
        self.slot = {}
        self.slots = []
        for t in self.types:
            for m in self.members[t]:
                self.slot[(t, m)] = len(self.slots)
                self.slots.append((t, m))
        names = {t: set() for t in self.types}
        for (rt, rid, _rel, st, sid, _sr) in tuples:
            names[rt].add(rid)
            names[st].add(sid)
        self.extra = {t: set() for t in self.types}
        self.names = {t: sorted(v) for t, v in names.items()}
        self.oid = {t: {n: i for i, n in enumerate(v)} for t, v in self.names.items()}

    def ensure(self, t, name):  # ids for objects that appear only in requests
        if name not in self.oid[t]:
            self.oid[t][name] = len(self.names[t])
            self.names[t].append(name)
        return self.oid[t][name]

    def items(self, queries):
        """queries: [(rtype, rid, perm, stype, sid, srel)] -> acl_item_t array"""
        it = np.zeros(len(queries), dtype=ITEM_DTYPE)
        for i, (rt, rid, pm, st, sid, sr) in enumerate(queries):
            it[i] = (self.tid[rt], self.members[rt].index(pm), self.ensure(rt, rid), self.tid[st],
                     self.members[st].index(sr) if sr else NO_RELATION, self.ensure(st, sid))
        return it


class CpuShard:
    def __init__(self, universe: Universe, rank: int, world: int):
        self.u, self.o = universe, universe.o
        self.rank, self.world = rank, world
        self.device = torch.device("cpu")
        self.owner = {t: fnv1a(t) % world if world > 1 else 0 for t in universe.types}
        self.frontier = []
        # reverse indices over the relationships this shard owns
        self.by_subject, self.by_subject_obj = {}, {}
        for (rt, rid, rel), subs in self.o.rows.items():
            if self.owner[rt] != rank:
                continue
            for (st, sid, sr) in subs:
                self.by_subject.setdefault((st, sid, sr), []).append((rt, rid, rel))
                self.by_subject_obj.setdefault((st, sid), []).append((rt, rid, rel))

    def owner_of_type(self, t):
        return self.owner[t]

    def grow_frontier(self):
        pass

    # ------------------------------------------------------------------ Check
    def _enc(self, t, name, member, level, req, flags=0):
        return (self.u.ensure(t, name), req, self.u.slot[(t, member)] | (level << 13) | flags, 0)

    def _dec(self, row):
        oid, req, meta, _ = (int(x) for x in row)
        t, m = self.u.slots[meta & 0x1FFF]
        return t, self.u.names[t][oid], m, (meta >> 13) & 63, req, meta

    def check_begin(self, items, has, err):
        it = items.numpy().view(ITEM_DTYPE)
        self.subjects = []
        self.frontier = []
        has.zero_()
        err.zero_()
        for i, q in enumerate(it):
            rt, st = int(q["resource_type"]), int(q["subject_type"])
            ok = rt < len(self.u.types) and st < len(self.u.types)
            if ok:
                tn, sn = self.u.types[rt], self.u.types[st]
                ok = int(q["permission"]) < len(self.u.members[tn]) and (
                    int(q["subject_relation"]) == NO_RELATION or int(q["subject_relation"]) < len(self.u.members[sn]))
            if not ok:
                err[i] = 2
                self.subjects.append(None)
                continue
            sr = "" if int(q["subject_relation"]) == NO_RELATION else self.u.members[sn][int(q["subject_relation"])]
            self.subjects.append((sn, self.u.names[sn][int(q["subject_id"])], sr))
            if self.owner[tn] == self.rank:
                self.frontier.append((tn, self.u.names[tn][int(q["resource_id"])], self.u.members[tn][int(q["permission"])], 1, i))

    def _children(self, t, name, member):
        mem = self.o.defs[t].members[member]
        if isinstance(mem, Relation):
            return [(st, sid, sr) for (st, sid, sr) in self.o._subjects(t, name, member) if sr]
        out = []

        def walk(e):
            if e[0] == "union":
                walk(e[1])
                walk(e[2])
            elif e[0] == "ref":
                out.append((t, name, e[1]))
            elif e[0] == "arrow":
                for (st, sid, _sr) in self.o._subjects(t, name, e[1]):
                    if e[2] in self.o.defs[st].members:
                        out.append((st, sid, e[2]))
        walk(mem.expr)
        return out

    def check_step(self, level, has, err, export):
        nxt, exports = [], []
        for (t, name, member, lv, req) in self.frontier:
            assert lv == level
            if has[req]:
                continue
            subj = self.subjects[req]
            mem = self.o.defs[t].members[member]
            if subj == (t, name, member) or (isinstance(mem, Relation) and subj in self.o._subjects(t, name, member)):
                has[req] = 1
                continue
            for (ct, cname, cm) in self._children(t, name, member):
                if lv + 1 > MAX_DEPTH:
                    err[req] = max(int(err[req]), 1)
                elif self.owner[ct] == self.rank:
                    nxt.append((ct, cname, cm, lv + 1, req))
                else:
                    exports.append(self._enc(ct, cname, cm, lv + 1, req))
        self.frontier = nxt
        self._last_exports = exports
        for i, row in enumerate(exports[:export.shape[0]]):
            export[i] = torch.tensor(row, dtype=torch.int64).to(torch.int32)
        return len(exports), 1 if nxt else 0, 0

    def check_step_by_dest(self, level, has, err, export, cap_per_dest):
        tmp = torch.zeros((max(1, export.shape[0]), 4), dtype=torch.int32)
        n, produced, ov = self.check_step(level, has, err, tmp[:0])  # count only; entries regrouped below
        rows = self._last_exports
        counts = [0] * self.world
        for row in rows:
            t, _m = self.u.slots[row[2] & 0x1FFF]
            d = self.owner[t]
            if counts[d] < cap_per_dest:
                export[d * cap_per_dest + counts[d]] = torch.tensor(row, dtype=torch.int64).to(torch.int32)
            counts[d] += 1
        return (max(counts) if counts else 0, produced, ov), counts

    def check_import(self, level, entries, n):
        for row in entries[:n].tolist():
            t, name, m, lv, req, _meta = self._dec(row)
            if self.owner[t] == self.rank:
                assert lv == level + 1
                self.frontier.append((t, name, m, lv, req))

    def check_finish(self, has, err, perm, errout):
        h, e = has.numpy().astype(bool), err.numpy()
        n = h.size
        perm[:n] = torch.from_numpy(np.where(h, 2, np.where(e != 0, 0, 1)).astype(np.uint8))
        errout[:n] = torch.from_numpy(np.where(h, 0, np.where(e == 1, 100, np.where(e == 2, 9, 0))).astype(np.int32))

    # ------------------------------------------------------------------ LookupResources
    def lookup_begin(self, rtype, perm, stype, srel, sids):
        self.target = (rtype, perm)
        self.lsubjects = [(stype, self.u.names[stype][int(s)], srel or "") for s in sids]
        self.visited = [set() for _ in sids]
        self.lfront = [("seed", i) for i in range(len(sids))]

    def lookup_words(self, rtype):
        return max(1, (len(self.u.names[rtype]) + 31) // 32)

    def _parents(self, t, name, member, native):
        out = []
        if native:  # computed usersets on the same object: run by the owner only
            for pn, pm in self.o.defs[t].members.items():
                if isinstance(pm, Permission) and self._refs(pm.expr, member):
                    out.append((t, name, pn))
        for (rt, rid, rel) in self.by_subject.get((t, name, member), []):  # userset subjects
            out.append((rt, rid, rel))
        for (rt, rid, rel) in self.by_subject_obj.get((t, name), []):  # arrows rel->member
            for pn, pm in self.o.defs[rt].members.items():
                if isinstance(pm, Permission) and self._arrows(pm.expr, rel, member):
                    out.append((rt, rid, pn))
        return out

    def _refs(self, e, name):
        return (e[0] == "ref" and e[1] == name) or (e[0] == "union" and (self._refs(e[1], name) or self._refs(e[2], name)))

    def _arrows(self, e, ts, comp):
        return (e[0] == "arrow" and e[1] == ts and e[2] == comp) or (e[0] == "union" and (self._arrows(e[1], ts, comp) or self._arrows(e[2], ts, comp)))

    def lookup_step(self, it, phase, export):
        if phase == 2:  # EXPAND
            nxt = []
            for ent in self.lfront:
                if ent[0] == "seed":
                    req = ent[1]
                    st, sid, sr = self.lsubjects[req]
                    if sr and self.owner[st] == self.rank:
                        nxt.append((st, sid, sr, 1, req))
                    for (rt, rid, rel) in self.by_subject.get((st, sid, sr), []):
                        nxt.append((rt, rid, rel, 1, req))
                else:
                    t, name, m, dist, req, native = ent
                    if dist < MAX_DEPTH:
                        nxt.extend((pt, pn, pm, dist + 1, req) for (pt, pn, pm) in self._parents(t, name, m, native))
            self.lfront = nxt
            return 0, 1 if nxt else 0, 0
        passed, exports = [], []
        for (t, name, m, dist, req) in self.lfront:
            assert self.owner[t] == self.rank
            if (t, name, m) in self.visited[req]:
                continue
            self.visited[req].add((t, name, m))
            passed.append((t, name, m, dist, req, True))
            if self.world > 1 and dist < MAX_DEPTH:
                exports.append(self._enc(t, name, m, dist, req, FOREIGN))
        self.lfront = passed
        for i, row in enumerate(exports[:export.shape[0]]):
            export[i] = torch.tensor(row, dtype=torch.int64).to(torch.int32)
        return len(exports), 1 if passed else 0, 0

    def lookup_import(self, it, entries, n):
        for row in entries[:n].tolist():
            t, name, m, dist, req, meta = self._dec(row)
            assert meta & FOREIGN and self.owner[t] != self.rank
            self.lfront.append((t, name, m, dist, req, False))

    def lookup_finish(self, bitmaps):
        bitmaps.zero_()
        rt, pm = self.target
        if self.owner[rt] != self.rank:
            return
        bm = np.zeros(tuple(bitmaps.shape), dtype=np.uint32)
        for i, vis in enumerate(self.visited):
            for (t, name, m) in vis:
                if t == rt and m == pm:
                    b = self.u.oid[t][name]
                    bm[i, b >> 5] |= np.uint32(1 << (b & 31))
        bitmaps.copy_(torch.from_numpy(bm.view(np.int32)))
