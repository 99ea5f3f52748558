// plan.cpp -- builds the HBM snapshot (CSR rows) and the frontier programs.  See plan.hpp.
#include "plan.hpp"

#include <algorithm>
#include <functional>

namespace acl {
namespace {

constexpr uint32_t kMaxOpsPerSlot = 256;
constexpr uint32_t kMaxDepth = 50;  // pkg/spicedb/spicedb.go:34

struct RelLayout {
    bool any = false;
    uint32_t meta_base = 0, nrows = 0, K = 0;  // meta_base in uint2 units
    std::vector<uint8_t> class_live;    // class has >= 1 live relationship
    std::vector<uint8_t> class_hashed;  // class is stored as hashed buckets (never enumerated)
};

inline uint32_t hash_bucket(uint32_t sid, uint32_t nb) { return (uint32_t)(((uint64_t)(sid * 0x9E3779B1u) * nb) >> 32); }
inline uint32_t buckets_for(uint32_t n) { return n <= 4 ? 1u : (n + 2) / 3; }

struct Flattener {
    const Schema &sc;
    const std::vector<RelLayout> &lay;  // [slot]
    std::vector<FwdOp> probes, makers, reflex;
    std::vector<int> stack;
    uint32_t max_d = 0;

    size_t nops() const { return probes.size() + makers.size() + reflex.size(); }
    void push_same(int target, uint32_t d) {
        FwdOp op{};
        op.flags = OP_PUSH_SAME;
        op.dlevel = d;
        op.key = (uint32_t)target;
        makers.push_back(op);
    }
    void row_op(uint32_t flags, int rel_slot, int k, uint32_t d, uint32_t key) {
        const RelLayout &l = lay[rel_slot];
        if (!l.any || !l.class_live[k]) return;  // empty class: nothing to probe or enumerate in this snapshot
        FwdOp op{};
        op.flags = flags;
        op.dlevel = d;
        op.meta_base = l.meta_base;
        op.nrows = l.nrows;
        op.K = l.K;
        op.k = (uint32_t)k;
        op.key = key;
        ((flags & (OP_ENUM | OP_PUSH_SAME)) ? makers : probes).push_back(op);
    }
    // state (type, member) entered at depth offset d
    void state(int type, int member, uint32_t d) {
        const Member &m = sc.defs[type].members[member];
        max_d = std::max(max_d, d);
        {  // the subject itself, when it is exactly this object#relation, is a member
            FwdOp op{};
            op.flags = OP_REFLEX;
            op.dlevel = d;
            op.key = (uint32_t)m.slot;
            reflex.push_back(op);
        }
        if (!m.is_permission) {
            const RelLayout &l = lay[m.slot];
            for (size_t k = 0; k < m.classes.size(); k++) {
                const SubjectClass &c = m.classes[k];
                if (c.srel == kNoRelation)
                    row_op(l.any && l.class_hashed[k] ? OP_PROBE_HASH : OP_PROBE, m.slot, (int)k, d, sc.subject_key(c.stype, kNoRelation));
                else
                    row_op(OP_PROBE | OP_ENUM, m.slot, (int)k, d, (uint32_t)sc.slot(c.stype, c.srel));
            }
            return;
        }
        stack.push_back(m.slot);
        expr(type, m.expr, d);
        stack.pop_back();
    }
    void expr(int type, const Node &n, uint32_t d) {
        const Definition &def = sc.defs[type];
        switch (n.kind) {
            case Node::kNil: break;
            case Node::kUnion:
                for (const Node &k : n.kids) expr(type, k, d);
                break;
            case Node::kRef: {  // computed userset: a dispatch on the same object, one level deeper
                int tm = def.find(n.a);
                int tslot = sc.slot(type, tm);
                bool recursive = std::find(stack.begin(), stack.end(), tslot) != stack.end();
                if (recursive || d + 1 > kMaxDepth || nops() >= kMaxOpsPerSlot) push_same(tslot, d);
                else state(type, tm, d + 1);
                break;
            }
            case Node::kArrow: {  // tuple-to-userset: children dispatched one level deeper
                int ts = def.find(n.a);
                const Member &rel = def.members[ts];
                for (size_t k = 0; k < rel.classes.size(); k++) {
                    int st = rel.classes[k].stype;
                    int tm = sc.defs[st].find(n.b);
                    if (tm < 0) continue;  // subject type lacks the computed relation: not dispatched
                    row_op(OP_ENUM, rel.slot, (int)k, d, (uint32_t)sc.slot(st, tm));
                }
                break;
            }
        }
    }
};

void collect(const Node &n, Node::Kind kind, std::vector<const Node *> *out) {
    if (n.kind == kind) out->push_back(&n);
    for (const Node &k : n.kids) collect(k, kind, out);
}

}  // namespace

void build_forward(Store &store, int64_t now, Snapshot *snap) {
    store.settle_all();
    const Schema &sc = store.schema();
    Snapshot s;
    s.revision = store.revision();
    store.expiry_window(now, &s.valid_lo, &s.valid_hi);
    s.nslots = (uint32_t)sc.nslots;
    s.ntypes = (uint32_t)sc.defs.size();
    for (size_t t = 0; t < sc.defs.size(); t++) {
        s.type_slot_base.push_back((uint32_t)sc.slot_base[t]);
        s.type_nmembers.push_back((uint32_t)sc.defs[t].members.size());
        s.type_nobjects.push_back(store.objects((int)t).count());
    }
    // relations that some arrow walks as a tupleset must stay enumerable (SORTED)
    std::vector<uint8_t> is_tupleset(sc.nslots, 0);
    for (const Definition &d : sc.defs)
        for (const Member &m : d.members) {
            if (!m.is_permission) continue;
            std::vector<const Node *> arrows;
            collect(m.expr, Node::kArrow, &arrows);
            for (const Node *a : arrows) is_tupleset[d.members[d.find(a->a)].slot] = 1;
        }
    std::vector<RelLayout> lay(sc.nslots);
    auto &tables = store.tables();
    std::vector<uint32_t> row;  // scratch: live subjects of one (object, class) sub-row
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        if (mem.is_permission) continue;
        RelLayout &l = lay[slot];
        l.K = (uint32_t)mem.classes.size();
        l.nrows = store.objects(t).count();
        l.class_live.assign(l.K, 0);
        l.class_hashed.assign(l.K, 0);
        size_t total = 0;
        for (uint32_t k = 0; k < l.K; k++) {
            const ClassTable &ct = tables[slot][k];
            l.class_hashed[k] = mem.classes[k].srel == kNoRelation && !is_tupleset[slot];
            if (ct.expiry.empty()) {
                if (!ct.keys.empty()) l.class_live[k] = 1;
                total += ct.keys.size();
            } else {
                for (uint64_t key : ct.keys)
                    if (store.live(ct, key, now)) { l.class_live[k] = 1; total++; }
            }
        }
        if (!total) continue;
        l.any = true;
        if (s.meta.size() % 4) s.meta.resize(s.meta.size() + 2, 0);  // 16-byte alignment: K == 2 rows load as one dwordx4
        l.meta_base = (uint32_t)(s.meta.size() / 2);
        const size_t nrow = (size_t)l.nrows * l.K;
        s.meta.resize(s.meta.size() + 2 * nrow, 0);
        uint32_t *meta = s.meta.data() + 2 * (size_t)l.meta_base;
        for (uint32_t k = 0; k < l.K; k++) {
            const ClassTable &ct = tables[slot][k];
            const bool filt = !ct.expiry.empty();
            size_t i = 0;
            const size_t nk = ct.keys.size();
            while (i < nk) {  // one resource at a time (keys ascend by resource, then subject)
                const uint32_t res = (uint32_t)(ct.keys[i] >> 32);
                row.clear();
                for (; i < nk && (uint32_t)(ct.keys[i] >> 32) == res; i++)
                    if (!filt || store.live(ct, ct.keys[i], now)) row.push_back((uint32_t)ct.keys[i]);
                if (row.empty()) continue;
                uint32_t *md = meta + 2 * ((size_t)res * l.K + k);
                if (l.class_hashed[k]) {
                    const uint32_t nb = buckets_for((uint32_t)row.size());
                    const uint32_t b0 = (uint32_t)(s.buckets.size() / 4);
                    s.buckets.resize(s.buckets.size() + 4 * (size_t)nb, 0xFFFFFFFFu);
                    uint32_t *bk = s.buckets.data() + 4 * (size_t)b0;
                    for (uint32_t sid : row) {
                        uint32_t b = hash_bucket(sid, nb);
                        for (;;) {
                            uint32_t *q = bk + 4 * (size_t)b;
                            int f = q[0] == 0xFFFFFFFFu ? 0 : q[1] == 0xFFFFFFFFu ? 1 : q[2] == 0xFFFFFFFFu ? 2 : q[3] == 0xFFFFFFFFu ? 3 : -1;
                            if (f >= 0) { q[f] = sid; break; }
                            b = b + 1 == nb ? 0 : b + 1;
                        }
                    }
                    md[0] = b0;
                    md[1] = b0 + nb;
                } else {
                    md[0] = (uint32_t)s.edges.size();
                    s.edges.insert(s.edges.end(), row.begin(), row.end());
                    md[1] = (uint32_t)s.edges.size();
                }
            }
        }
    }
    s.nedges = 0;
    for (const auto &slot : tables)
        for (const auto &ct : slot) s.nedges += ct.keys.size();
    if (s.meta.empty()) s.meta.assign(4, 0);
    if (s.edges.empty()) s.edges.push_back(0);
    if (s.buckets.empty()) s.buckets.assign(4, 0xFFFFFFFFu);
    // programs
    s.progs.resize(sc.nslots);
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        Flattener f{sc, lay, {}, {}, {}, {}, 0};
        f.state(t, m, 0);
        SlotProg p{};
        p.first = (uint32_t)s.ops.size();
        p.n_probe = (uint32_t)f.probes.size();
        p.n_main = (uint32_t)(f.probes.size() + f.makers.size());
        p.n_total = (uint32_t)f.nops();
        p.max_dlevel = f.max_d;
        s.ops.insert(s.ops.end(), f.probes.begin(), f.probes.end());
        s.ops.insert(s.ops.end(), f.makers.begin(), f.makers.end());
        s.ops.insert(s.ops.end(), f.reflex.begin(), f.reflex.end());
        s.progs[slot] = p;
    }
    if (s.ops.empty()) s.ops.push_back(FwdOp{});
    *snap = std::move(s);
}

void build_reverse(Store &store, int64_t now, Snapshot *snap) {
    const Schema &sc = store.schema();
    Snapshot &s = *snap;
    auto &tables = store.tables();
    s.roff.clear();
    s.redges.clear();
    s.rops.clear();
    // reverse rows per (relation slot, class): subject id -> sorted resource ids
    struct RevLayout { bool any = false; uint32_t roff_base = 0, nrows = 0; };
    std::vector<std::vector<RevLayout>> rl(sc.nslots);
    std::vector<uint32_t> cursor;
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        rl[slot].resize(mem.classes.size());
        for (size_t k = 0; k < mem.classes.size(); k++) {
            const ClassTable &ct = tables[slot][k];
            const bool filt = !ct.expiry.empty();
            const uint32_t ns = store.objects(mem.classes[k].stype).count();
            size_t total = 0;
            for (uint64_t key : ct.keys)
                if (!filt || store.live(ct, key, now)) total++;
            if (!total) continue;
            RevLayout &l = rl[slot][k];
            l.any = true;
            l.nrows = ns;
            l.roff_base = (uint32_t)s.roff.size();
            s.roff.resize(s.roff.size() + ns + 1, 0);
            uint32_t *ro = s.roff.data() + l.roff_base;
            for (uint64_t key : ct.keys)
                if (!filt || store.live(ct, key, now)) ro[(uint32_t)key]++;
            uint32_t run = (uint32_t)s.redges.size();
            for (uint32_t i = 0; i < ns; i++) {
                uint32_t c = ro[i];
                ro[i] = run;
                run += c;
            }
            ro[ns] = run;
            s.redges.resize(run);
            cursor.assign(ro, ro + ns);
            for (uint64_t key : ct.keys)  // keys ascend by resource => each reverse row ascends by resource
                if (!filt || store.live(ct, key, now)) s.redges[cursor[(uint32_t)key]++] = (uint32_t)(key >> 32);
        }
    }
    if (s.roff.empty()) s.roff.push_back(0);
    if (s.redges.empty()) s.redges.push_back(0);
    auto enum_op = [&](int rel_slot, size_t k, int target) {
        const RevLayout &l = rl[rel_slot][k];
        if (!l.any) return;
        RevOp op{};
        op.flags = OP_ENUM;
        op.roff_base = l.roff_base;
        op.nrows = l.nrows;
        op.target = (uint32_t)target;
        s.rops.push_back(op);
    };
    // parents of a true state X = (t, m)
    s.rprogs.assign(sc.nslots, RevProg{0, 0});
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const std::string &xname = sc.defs[t].members[m].name;
        RevProg p;
        p.first = (uint32_t)s.rops.size();
        for (size_t t2 = 0; t2 < sc.defs.size(); t2++) {
            const Definition &d2 = sc.defs[t2];
            for (const Member &m2 : d2.members) {
                if (!m2.is_permission) {
                    // userset subjects `t:id#m` stored on relation m2
                    for (size_t k = 0; k < m2.classes.size(); k++)
                        if (m2.classes[k].stype == t && m2.classes[k].srel == m) enum_op(m2.slot, k, m2.slot);
                    continue;
                }
                std::vector<const Node *> refs, arrows;
                collect(m2.expr, Node::kRef, &refs);
                collect(m2.expr, Node::kArrow, &arrows);
                if ((int)t2 == t)
                    for (const Node *r : refs)
                        if (r->a == xname) {
                            RevOp op{};
                            op.flags = OP_PUSH_SAME;
                            op.target = (uint32_t)m2.slot;
                            s.rops.push_back(op);
                            break;
                        }
                for (const Node *a : arrows) {
                    if (a->b != xname) continue;
                    const Member &ts = d2.members[d2.find(a->a)];
                    for (size_t k = 0; k < ts.classes.size(); k++)
                        if (ts.classes[k].stype == t) enum_op(ts.slot, k, m2.slot);
                }
            }
        }
        p.n = (uint32_t)s.rops.size() - p.first;
        s.rprogs[slot] = p;
    }
    // seeds for a subject key
    s.rseeds.assign(sc.nkeys(), RevProg{0, 0});
    for (uint32_t key = 0; key < sc.nkeys(); key++) {
        int st, sr;
        if (key < (uint32_t)sc.nslots) {
            st = sc.slot_owner[key].first;
            sr = sc.slot_owner[key].second;
        } else {
            st = (int)key - sc.nslots;
            sr = kNoRelation;
        }
        RevProg p;
        p.first = (uint32_t)s.rops.size();
        if (sr != kNoRelation) {  // reflexive: `t:id#m` is a member of t:id#m
            RevOp op{};
            op.flags = OP_PUSH_SAME;
            op.target = key;
            s.rops.push_back(op);
        }
        for (int slot = 0; slot < sc.nslots; slot++) {
            auto [t2, m2] = sc.slot_owner[slot];
            const Member &mem = sc.defs[t2].members[m2];
            for (size_t k = 0; k < mem.classes.size(); k++)
                if (mem.classes[k].stype == st && mem.classes[k].srel == sr) enum_op(slot, k, slot);
        }
        p.n = (uint32_t)s.rops.size() - p.first;
        s.rseeds[key] = p;
    }
    if (s.rops.empty()) s.rops.push_back(RevOp{});
    s.slot_bit_base.assign(sc.nslots + 1, 0);
    s.slot_nobjects.assign(sc.nslots, 0);
    uint64_t bits = 0;
    for (int slot = 0; slot < sc.nslots; slot++) {
        s.slot_bit_base[slot] = (uint32_t)bits;
        s.slot_nobjects[slot] = store.objects(sc.slot_owner[slot].first).count();
        bits += ((uint64_t)s.slot_nobjects[slot] + 31) / 32 * 32;
    }
    s.slot_bit_base[sc.nslots] = (uint32_t)bits;
    s.visited_bits = bits;
    s.has_reverse = true;
}

}  // namespace acl
