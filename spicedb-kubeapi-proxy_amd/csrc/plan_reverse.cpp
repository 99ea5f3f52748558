// plan_reverse.cpp -- reverse rows + parent programs for LookupResources (reference pkg/authz/lookups.go:49-65).
// A state X = (type, relation|permission, id) being TRUE for the subject makes its parents true:
//   - permissions on the same object whose rewrite references X's member (computed userset),
//   - permissions with an arrow `a->b` (b == X's member) on every resource whose tupleset `a` holds X's object,
//   - relations that store `type:id#member` as a userset subject.
// The walk starts from the relationships that name the subject directly (plus the subject itself when it
// carries a relation).  See plan.hpp.
#include <algorithm>

#include "plan.hpp"

namespace acl {
namespace {

// References / arrows in POSITIVE positions: a rewrite with `&` / `-` is walked backwards as a superset -- X true makes the parent a CANDIDATE
// when X occurs anywhere but under the subtracted operand of an exclusion (every true value of `a - b`, `a & b` has a true positive operand);
// the engine then runs the forward Check over the candidates (engine.cpp lookup_batch, Snapshot::slot_nonmono).
void collect(const Node &n, Node::Kind kind, std::vector<const Node *> *out) {
    // (a.all(b) true needs a true b on EVERY a: any true b makes the parent a candidate, as for a->b)
    if (n.kind == kind || (kind == Node::kArrow && n.kind == Node::kArrowAll)) out->push_back(&n);
    if (n.kind == Node::kExclude) {
        collect(n.kids[0], kind, out);
        return;
    }
    for (const Node &k : n.kids) collect(k, kind, out);
}

}  // namespace

void build_reverse(Store &store, int64_t now, Snapshot *snap, ShardSpec shard) {
    const Schema &sc = store.schema();
    Snapshot &s = *snap;
    std::vector<uint32_t> type_owner;
    for (const Definition &d : sc.defs) type_owner.push_back(shard_of_type(d.name, shard.world));
    auto &tables = store.tables();
    s.rmeta.clear();
    s.redges.clear();
    s.redge_cap.clear();
    s.rops.clear();
    // reverse rows per (relation slot, class): subject id -> resource ids; one {start, end} descriptor per subject,
    // sized with headroom so that writes naming new subjects can be patched in (patch_reverse)
    using RevLayout = Snapshot::RevLayout;
    std::vector<std::vector<RevLayout>> rl(sc.nslots);
    std::vector<uint32_t> cursor;
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        rl[slot].resize(mem.classes.size());
        for (size_t k = 0; k < mem.classes.size(); k++) {
            const ClassTable &ct = tables[slot][k];
            // `any` is a property of the whole graph (it decides which shards a true state must be sent to);
            // the rows themselves are only built on the shard that owns the relation's type
            // every declared class counts as live, with or without relationships (as in build_forward: the first relationship of a class
            // must be a patch, not a reason to rebuild the parents' programs)
            if (type_owner[t] != shard.rank) {
                rl[slot][k].any = true;
                continue;
            }
            const bool filt = !ct.expiry.empty();
            const uint32_t ns = mem.classes[k].wildcard ? store.wildcard_id(mem.classes[k].stype) + 1 : with_headroom(store.objects(mem.classes[k].stype).count());
            RevLayout &l = rl[slot][k];
            l.any = true;
            l.nrows = ns;
            l.base = (uint32_t)(s.rmeta.size() / 2);
            s.rmeta.resize(s.rmeta.size() + 2 * (size_t)ns, 0);
            cursor.assign((size_t)ns + 1, 0);
            for (uint64_t key : ct.keys)
                if ((uint32_t)key < ns && (!filt || store.live(ct, key, now))) cursor[(uint32_t)key + 1]++;
            uint32_t *rm = s.rmeta.data() + 2 * (size_t)l.base;
            uint32_t run = (uint32_t)s.redges.size();
            for (uint32_t i = 0; i < ns; i++) {
                const uint32_t c = cursor[i + 1];
                rm[2 * i] = run;
                cursor[i] = run;
                run += c;
                rm[2 * i + 1] = run;
            }
            s.redges.resize(run);
            for (uint64_t key : ct.keys)  // keys ascend by resource => each reverse row ascends by resource
                if ((uint32_t)key < ns && (!filt || store.live(ct, key, now))) s.redges[cursor[(uint32_t)key]++] = (uint32_t)(key >> 32);
        }
    }
    if (s.rmeta.empty()) s.rmeta.assign(2, 0);
    if (s.redges.empty()) s.redges.push_back(0);
    bool remote = false;  // set by enum_op when a live parent row set belongs to another shard
    uint64_t remote_shards = 0;  // ... which ones (shards beyond 63 are reached by the all-gather form only)
    auto enum_op = [&](int rel_slot, size_t k, int target, uint32_t wild_id = 0xFFFFFFFFu) {
        const RevLayout &l = rl[rel_slot][k];
        if (!l.any) return;
        if (type_owner[sc.slot_owner[rel_slot].first] != shard.rank) {
            remote = true;
            if (type_owner[sc.slot_owner[rel_slot].first] < 64) remote_shards |= 1ull << type_owner[sc.slot_owner[rel_slot].first];
            return;
        }
        RevOp op{};
        op.flags = OP_ENUM;
        op.roff_base = l.base;
        op.nrows = l.nrows;
        if (wild_id != 0xFFFFFFFFu) {  // `T:*`: the wildcard subject's row, whatever the seed's id
            op.flags |= OP_WILD;
            op.roff_base = l.base + wild_id;
        }
        op.target = (uint32_t)target;
        s.rops.push_back(op);
    };
    // parents of a true state X = (t, m)
    s.rprogs.assign(sc.nslots, RevProg{0, 0});
    s.rdest.assign(sc.nslots, 0);
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const std::string &xname = sc.defs[t].members[m].name;
        RevProg p;
        p.first = (uint32_t)s.rops.size();
        remote = false;
        remote_shards = 0;
        const bool mine = type_owner[t] == shard.rank;  // computed usersets stay on the object: only its owner runs them
        for (size_t t2 = 0; t2 < sc.defs.size(); t2++) {
            const Definition &d2 = sc.defs[t2];
            for (const Member &m2 : d2.members) {
                if (!m2.is_permission) {
                    // userset subjects `t:id#m` stored on relation m2
                    for (size_t k = 0; k < m2.classes.size(); k++)
                        if (m2.classes[k].stype == t && m2.classes[k].srel == m && !m2.classes[k].wildcard) enum_op(m2.slot, k, m2.slot);
                    continue;
                }
                std::vector<const Node *> refs, arrows;
                collect(m2.expr, Node::kRef, &refs);
                collect(m2.expr, Node::kArrow, &arrows);
                if ((int)t2 == t && mine)
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
        p.n = ((uint32_t)s.rops.size() - p.first) | (remote && mine ? kRevRemoteBit : 0u);
        s.rprogs[slot] = p;
        s.rdest[slot] = remote && mine ? remote_shards : 0;
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
        if (sr != kNoRelation && type_owner[st] == shard.rank) {  // reflexive: `t:id#m` is a member of t:id#m
            RevOp op{};
            op.flags = OP_PUSH_SAME;
            op.target = key;
            s.rops.push_back(op);
        }
        for (int slot = 0; slot < sc.nslots; slot++) {
            auto [t2, m2] = sc.slot_owner[slot];
            const Member &mem = sc.defs[t2].members[m2];
            for (size_t k = 0; k < mem.classes.size(); k++)
                if (mem.classes[k].stype == st && mem.classes[k].srel == sr)  // (a plain subject is also covered by every `st:*` relationship)
                    enum_op(slot, k, slot, mem.classes[k].wildcard ? store.wildcard_id(st) : 0xFFFFFFFFu);
        }
        p.n = (uint32_t)s.rops.size() - p.first;
        s.rseeds[key] = p;
    }
    // Seed ops whose target slot has no other producer anywhere: their children need no visited bits (kernels.hip k_rev_local) -- in the
    // reference's own schema that is every relation a user is named on directly (pod#viewer, pod#creator, namespace#viewer, ...): two thirds
    // of the atomics of a list-pods lookup.  (Sharded graphs keep their bits: other shards may produce the same state.)
    if (shard.world == 1) {
        std::vector<uint32_t> produced(sc.nslots, 0);
        for (const RevProg &p : s.rprogs)
            for (uint32_t j = 0; j < (p.n & ~kRevRemoteBit); j++) produced[s.rops[p.first + j].target]++;
        for (const RevProg &p : s.rseeds) {
            for (uint32_t j = 0; j < p.n; j++) {
                const uint32_t tgt = s.rops[p.first + j].target;
                uint32_t same = 0;
                for (uint32_t k = 0; k < p.n; k++) same += s.rops[p.first + k].target == tgt ? 1u : 0u;
                if (!produced[tgt] && same == 1) s.rops[p.first + j].flags |= OP_NOMARK;
            }
        }
    }
    // which slots are sinks as a lookup's result: the slot cannot be reached again from its own parents (group#member reaches itself through nested groups;
    // `edit` under `view = viewer + edit`, or namespace#view under pod#view = ... + namespace->view, do not)
    // ... and, per result slot, the slots that can lead to it (the parent graph walked backwards from t)
    {
        constexpr uint32_t kWords = 256 / 32;  // (kernels.hpp kRevUsefulWords: k_rev_local's LDS copy holds 256 slots; larger schemas take the level loop)
        std::vector<std::vector<int>> producers(sc.nslots);  // producers[y]: slots x with an op x -> y
        for (int x = 0; x < sc.nslots; x++) {
            const RevProg &p = s.rprogs[x];
            for (uint32_t j = 0; j < (p.n & ~kRevRemoteBit); j++) producers[s.rops[p.first + j].target].push_back(x);
        }
        s.rev_useful.assign((size_t)sc.nslots * kWords, sc.nslots <= 256 ? 0u : 0xFFFFFFFFu);  // (beyond 256 slots nothing is pruned -- and k_rev_local is not used)
        for (int t = 0; t < sc.nslots && sc.nslots <= 256; t++) {
            uint32_t *row = s.rev_useful.data() + (size_t)t * kWords;
            std::vector<int> stack{t};
            row[t >> 5] |= 1u << (t & 31);
            while (!stack.empty()) {
                const int y = stack.back();
                stack.pop_back();
                for (int x : producers[y])
                    if (!((row[x >> 5] >> (x & 31)) & 1u)) {
                        row[x >> 5] |= 1u << (x & 31);
                        stack.push_back(x);
                    }
            }
        }
    }
    s.rev_sink.assign(sc.nslots, 0);
    for (int t = 0; t < sc.nslots; t++) {
        std::vector<uint8_t> seen(sc.nslots, 0);
        std::vector<int> stack{t};
        bool back = false;
        while (!stack.empty() && !back) {
            const int v = stack.back();
            stack.pop_back();
            const RevProg &p = s.rprogs[v];
            for (uint32_t j = 0; j < (p.n & ~kRevRemoteBit) && !back; j++) {
                const int w = (int)s.rops[p.first + j].target;
                if (w == t) back = true;
                else if (!seen[w]) {
                    seen[w] = 1;
                    stack.push_back(w);
                }
            }
        }
        s.rev_sink[t] = back ? 0 : 1;
    }
    if (s.rops.empty()) s.rops.push_back(RevOp{});
    s.slot_bit_base.assign(sc.nslots + 1, 0);
    s.slot_nobjects.assign(sc.nslots, 0);
    uint64_t bits = 0;
    for (int slot = 0; slot < sc.nslots; slot++) {
        s.slot_bit_base[slot] = (uint32_t)bits;
        s.slot_nobjects[slot] = with_headroom(store.objects(sc.slot_owner[slot].first).count());
        bits += ((uint64_t)s.slot_nobjects[slot] + 31) / 32 * 32;
    }
    s.slot_bit_base[sc.nslots] = (uint32_t)bits;
    s.visited_bits = bits;
    s.rlay = std::move(rl);
    s.has_reverse = true;
}


bool patch_reverse(Store &store, int64_t now, uint64_t from_revision, Snapshot *snap, ShardSpec shard, std::vector<Patch> *patches) {
    Snapshot &s = *snap;
    if (!s.has_reverse || s.rlay.empty()) return false;
    std::vector<Store::Change> ch;
    if (!store.raw_changes_since(from_revision, &ch)) return false;
    expiry_crossings(store, s.patch_lo, s.patch_hi, now, &ch);  // (the window patch_forward started from)
    const Schema &sc = store.schema();
    // new objects must fit the visited bitmaps and the descriptor tables
    for (int slot = 0; slot < sc.nslots; slot++)
        if (store.objects(sc.slot_owner[slot].first).count() > s.slot_nobjects[slot]) return false;
    std::sort(ch.begin(), ch.end(), [](const Store::Change &a, const Store::Change &b) {
        return a.slot != b.slot ? a.slot < b.slot : a.cls != b.cls ? a.cls < b.cls : a.key < b.key;
    });
    auto &tables = store.tables();
    for (const Store::Change &c : ch) {  // first pass: nothing is modified before we know everything is patchable
        const auto &l = s.rlay[c.slot][c.cls];
        const ClassTable &ct = tables[c.slot][c.cls];
        const bool want = ct.contains(c.key) && store.live(ct, c.key, now);
        if (!l.any && want) return false;  // (cannot happen: every declared class has reverse rows)
        if (l.any && s.type_owner[sc.slot_owner[c.slot].first] == shard.rank && (uint32_t)c.key >= l.nrows) return false;
    }
    for (size_t i = 0; i < ch.size(); i++) {
        const Store::Change &c = ch[i];
        if (i && ch[i - 1].slot == c.slot && ch[i - 1].cls == c.cls && ch[i - 1].key == c.key) continue;
        if (s.type_owner[sc.slot_owner[c.slot].first] != shard.rank) continue;
        const auto &l = s.rlay[c.slot][c.cls];
        if (!l.any) continue;
        const ClassTable &ct = tables[c.slot][c.cls];
        const bool want = ct.contains(c.key) && store.live(ct, c.key, now);
        const uint32_t res = (uint32_t)(c.key >> 32), sid = (uint32_t)c.key;
        uint32_t *md = s.rmeta.data() + 2 * ((size_t)l.base + sid);
        const uint32_t a = md[0], b = md[1];
        const auto first = s.redges.begin() + a, last = s.redges.begin() + b;
        const auto it = std::lower_bound(first, last, res);
        const bool have = it != last && *it == res;
        if (want == have) continue;
        if (want) {
            const uint32_t len = b - a, ipos = (uint32_t)(it - first);
            auto capit = s.redge_cap.find(a);
            const uint32_t cap = capit == s.redge_cap.end() ? len : capit->second;
            if (len && len + 1 <= cap) {  // the row was moved before and has room: insert in place
                std::copy_backward(s.redges.begin() + a + ipos, s.redges.begin() + b, s.redges.begin() + b + 1);
                s.redges[a + ipos] = res;
                md[1] = b + 1;
                patches->push_back(Patch{Patch::REDGES, (size_t)a + ipos, (size_t)(len + 1 - ipos)});
            } else {  // move the row to the end with the resource inserted in order and room for half as many again
                std::vector<uint32_t> row(first, last);
                row.insert(row.begin() + ipos, res);
                const uint32_t start = (uint32_t)s.redges.size(), ncap = std::max<uint32_t>(4, (len + 1) + (len + 1) / 2);
                s.redges.insert(s.redges.end(), row.begin(), row.end());
                s.redges.resize((size_t)start + ncap, 0u);
                if (capit != s.redge_cap.end()) s.redge_cap.erase(capit);
                s.redge_cap[start] = ncap;
                md = s.rmeta.data() + 2 * ((size_t)l.base + sid);
                md[0] = start;
                md[1] = start + len + 1;
                s.garbage_words += cap;
                patches->push_back(Patch{Patch::REDGES, start, (size_t)(len + 1)});
            }
        } else {  // shrink in place
            const uint32_t pos = (uint32_t)(it - s.redges.begin());
            std::copy(s.redges.begin() + pos + 1, s.redges.begin() + b, s.redges.begin() + pos);
            md[1]--;
            s.garbage_words++;
            if (md[1] > md[0]) patches->push_back(Patch{Patch::REDGES, md[0], (size_t)(md[1] - md[0])});
        }
        patches->push_back(Patch{Patch::RMETA, (size_t)(md - s.rmeta.data()), 2});
    }
    return true;
}

}  // namespace acl
