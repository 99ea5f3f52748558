// plan.cpp -- builds the forward HBM snapshot (sorted rows, subject-indexed hashed rows) and the frontier
// programs.  See plan.hpp.
#include "plan.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace acl {
namespace {

constexpr uint32_t kMaxOpsPerSlot = 256;
constexpr uint32_t kMaxDepth = 50;  // pkg/spicedb/spicedb.go:34


constexpr uint32_t kEmpty = 0xFFFFFFFFu;
// ---- hashed rows (plan.hpp): seeded single-choice where a seed can be found, two-choice (cuckoo) otherwise.  A row is named by its descriptor
// {first bucket, y = buckets | two-choice bit | seed}.
// two-choice insertion with random-walk eviction; false when the row is too tight (the caller gives it more buckets)
bool cuckoo_insert(uint32_t *row, uint32_t y, uint32_t id) {
    uint32_t cur = id;
    for (uint32_t kick = 0; kick < 512; kick++) {
        const uint32_t h1 = hrow_bucket(cur, y), h2 = hrow_bucket2(cur, y, h1);
        for (uint32_t b : {h1, h2})
            for (int k = 0; k < 4; k++)
                if (row[4 * (size_t)b + k] == kEmpty) {
                    row[4 * (size_t)b + k] = cur;
                    return true;
                }
        const uint32_t b = (kick & 1u) ? h2 : h1;
        const uint32_t k = (cur * 2654435761u >> 13 ^ kick) & 3u;
        std::swap(cur, row[4 * (size_t)b + k]);
    }
    // undo is not needed: the caller discards the row
    return false;
}
bool hashed_row_has(const uint32_t *row, uint32_t y, uint32_t id) {
    if (!hrow_nb(y)) return false;
    const uint32_t h1 = hrow_bucket(id, y);
    for (int k = 0; k < 4; k++)
        if (row[4 * (size_t)h1 + k] == id) return true;
    if (!(y & kRowTwoBit)) return false;
    const uint32_t h2 = hrow_bucket2(id, y, h1);
    for (int k = 0; k < 4; k++)
        if (row[4 * (size_t)h2 + k] == id) return true;
    return false;
}
inline uint32_t buckets_for(uint32_t n) { return n <= 4 ? 1u : (n + 2) / 3; }  // load <= 0.75 (<= 1.0 for a single bucket)
// A/B knob (ACL_SEEDED_ROWS=0: every row two-choice, as in rounds 1-4)
bool seeded_rows() {
    static const bool on = [] {
        const char *e = getenv("ACL_SEEDED_ROWS");
        return !(e && atoi(e) == 0);
    }();
    return on;
}
constexpr uint32_t kSeeds = 256;
// Places `ids` into `row` (nb buckets, all empty on entry) single-choice under the first seed that leaves no bucket with more than four ids;
// returns the seed or -1.  fill: scratch of nb bytes.
int seeded_place(uint32_t *row, uint32_t nb, const uint32_t *ids, size_t n, std::vector<uint8_t> &fill) {
    for (uint32_t seed = 0; seed < kSeeds; seed++) {
        const uint32_t y = hrow_pack(nb, seed, false);
        fill.assign(nb, 0);
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++) ok = ++fill[hrow_bucket(ids[i], y)] <= 4;
        if (!ok) continue;
        fill.assign(nb, 0);
        for (size_t i = 0; i < n; i++) {
            const uint32_t b = hrow_bucket(ids[i], y);
            row[4 * (size_t)b + fill[b]++] = ids[i];
        }
        return (int)seed;
    }
    return -1;
}
// Fills `row` (nb buckets) with `ids`: single-choice when a seed is found, else two-choice; returns the descriptor's y or 0 when the ids do not fit nb buckets
uint32_t place_row(uint32_t *row, uint32_t nb, const uint32_t *ids, size_t n, bool try_seeded, std::vector<uint8_t> &fill) {
    std::fill(row, row + 4 * (size_t)nb, kEmpty);
    if (try_seeded && (uint64_t)n <= 4ull * nb) {
        const int seed = seeded_place(row, nb, ids, n, fill);
        if (seed >= 0) return hrow_pack(nb, (uint32_t)seed, false);
    }
    if (buckets_for((uint32_t)n) > nb) return 0;
    const uint32_t y = hrow_pack(nb, 0, true);
    for (size_t i = 0; i < n; i++)
        if (!cuckoo_insert(row, y, ids[i])) return 0;
    return y;
}
// builds the row of `ids` at the end of `buckets`; returns {first bucket, y}.  Seeded single-choice at the smallest bucket count that finds a seed
// between load 0.75 and load 1/3 (min_buckets: the patcher's room to grow); beyond that two-choice at load <= 0.75.
std::pair<uint32_t, uint32_t> append_hashed_row(std::vector<uint32_t> &buckets, const uint32_t *ids, size_t n, uint32_t min_buckets) {
    const uint32_t b0 = (uint32_t)(buckets.size() / 4);
    if (!n && !min_buckets) return {b0, 0u};
    static thread_local std::vector<uint8_t> fill;
    const uint32_t nb0 = std::max(min_buckets, buckets_for((uint32_t)n));
    if ((uint64_t)nb0 + nb0 / 4 + 1 > kRowNbMask) throw std::runtime_error("a subject's hashed row exceeds 2^23 buckets");
    if (seeded_rows()) {
        const uint32_t nb_max = std::max<uint32_t>(nb0, (uint32_t)std::min<uint64_t>(((uint64_t)n * 3 + 3) / 4, kRowNbMask));  // load >= 1/3
        for (uint32_t nb = nb0; nb <= nb_max; nb += std::max(1u, nb / 8)) {
            buckets.resize(4 * (size_t)(b0 + nb));
            uint32_t *row = buckets.data() + 4 * (size_t)b0;
            std::fill(row, row + 4 * (size_t)nb, kEmpty);
            const int seed = seeded_place(row, nb, ids, n, fill);
            if (seed >= 0) return {b0, hrow_pack(nb, (uint32_t)seed, false)};
        }
    }
    for (uint32_t nb = nb0;; nb += std::max(1u, nb / 4)) {
        if (nb > kRowNbMask) throw std::runtime_error("a subject's hashed row exceeds 2^23 buckets");
        buckets.resize(4 * (size_t)(b0 + nb));
        const uint32_t y = place_row(buckets.data() + 4 * (size_t)b0, nb, ids, n, false, fill);
        if (y) return {b0, y};
    }
}

struct Flattener {
    const Schema &sc;
    const std::vector<RelLayout> &lay;  // [slot]
    const std::vector<uint32_t> &type_owner;
    uint32_t rank;
    const Store *store = nullptr;  // wildcard subject ids
    Flattener(const Schema &sc_, const std::vector<RelLayout> &lay_, const std::vector<uint32_t> &owner_, uint32_t rank_, const Store *store_)
        : sc(sc_), lay(lay_), type_owner(owner_), rank(rank_), store(store_) {}
    std::vector<FwdOp> probes, makers, reflex;
    std::vector<int> stack;
    uint32_t max_d = 0;
    // combine programs (rewrites with `&` / `-`): every op reports into a LEAF (0 = the state's own result cell)
    uint32_t cur_leaf = 0;
    std::vector<uint32_t> leaf_maxd{0};  // [leaf] deepest inlined dispatch offset
    std::vector<uint32_t> tokens;        // postfix boolean program over the leaves
    bool combine = false;

    size_t nops() const { return probes.size() + makers.size() + reflex.size(); }
    void note_depth(uint32_t d) {
        max_d = std::max(max_d, d);
        leaf_maxd[cur_leaf] = std::max(leaf_maxd[cur_leaf], d);
    }
    void push_same(int target, uint32_t d) {
        FwdOp op{};
        op.flags = OP_PUSH_SAME;
        op.dlevel = d;
        op.key = (uint32_t)target;
        op.leaf = cur_leaf;
        makers.push_back(op);
    }
    void row_op(uint32_t flags, int rel_slot, int k, uint32_t d, uint32_t key, uint32_t wild_id = 0xFFFFFFFFu) {
        const RelLayout &l = lay[rel_slot];
        const ClassLayout &c = l.cls[k];
        if (!c.live) return;  // rows held by another shard and referenced from here only through exports (sharded graph)
        FwdOp op{};
        op.dlevel = d;
        op.key = key;
        op.leaf = cur_leaf;
        if (c.hashed) {  // only ever asked for membership
            op.flags = OP_PROBE_HASH;
            op.base = c.smeta_base;
            op.nrows = c.nsubjects;
            if (wild_id != 0xFFFFFFFFu) {  // `T:*`: the row of the wildcard subject, whoever asks
                op.flags |= OP_WILD;
                op.K = wild_id;
            }
        } else {
            op.flags = flags;
            op.base = l.meta_base;
            op.nrows = l.nrows;
            op.K = l.Ks;
            op.k = c.ks;
        }
        ((op.flags & (OP_ENUM | OP_PUSH_SAME)) ? makers : probes).push_back(op);
    }
    // state (type, member) entered at depth offset d
    void state(int type, int member, uint32_t d) {
        const Member &m = sc.defs[type].members[member];
        note_depth(d);
        {  // the subject itself, when it is exactly this object#relation, is a member
            FwdOp op{};
            op.flags = OP_REFLEX;
            op.dlevel = d;
            op.key = (uint32_t)m.slot;
            op.leaf = cur_leaf;
            reflex.push_back(op);
        }
        if (!m.is_permission) {
            for (size_t k = 0; k < m.classes.size(); k++) {
                const SubjectClass &c = m.classes[k];
                if (c.wildcard) row_op(OP_PROBE, m.slot, (int)k, d, sc.subject_key(c.stype, kNoRelation), store->wildcard_id(c.stype));
                else if (c.srel == kNoRelation) row_op(OP_PROBE, m.slot, (int)k, d, sc.subject_key(c.stype, kNoRelation));
                else  // leaf flags are only computed (and only trusted) for children whose rows this shard holds
                    row_op(OP_PROBE | OP_ENUM | (type_owner[c.stype] == rank ? (uint32_t)OP_LEAFBIT : 0u), m.slot, (int)k, d,
                           (uint32_t)sc.slot(c.stype, c.srel));
            }
            return;
        }
        stack.push_back(m.slot);
        if (d == 0 && !m.expr.monotone()) root(type, m.expr);
        else expr(type, m.expr, d);
        stack.pop_back();
    }
    // ---- combine programs.  The state's value = OR(direct ops, the boolean program's result).  A union at the root keeps its monotone
    // operands direct (they answer the state's own cell, exactly as in a monotone program); everything under a `&` or `-` is cut into
    // leaves: maximal union-only sub-expressions, each flattened like a monotone program but reporting into its own cell.
    uint32_t new_leaf() {
        leaf_maxd.push_back(0);
        if (leaf_maxd.size() - 1 > kMaxLeaves) throw std::runtime_error("schema: a permission combines more than " + std::to_string(kMaxLeaves) + " operands under `&` / `-`");
        return (uint32_t)leaf_maxd.size() - 1;
    }
    void leaf_of(int type, const std::vector<const Node *> &parts) {  // one leaf for a union of monotone operands
        const uint32_t saved = cur_leaf;
        cur_leaf = new_leaf();
        for (const Node *n : parts) expr(type, *n, 0);
        tokens.push_back(BX_LEAF | cur_leaf);
        cur_leaf = saved;
    }
    void build(int type, const Node &n) {
        if (n.monotone()) {
            leaf_of(type, {&n});
            return;
        }
        switch (n.kind) {
            case Node::kUnion: {
                std::vector<const Node *> mono;
                uint32_t parts = 0;
                for (const Node &k : n.kids)
                    if (k.monotone()) mono.push_back(&k);
                if (!mono.empty()) {
                    leaf_of(type, mono);
                    parts++;
                }
                for (const Node &k : n.kids)
                    if (!k.monotone()) {
                        build(type, k);
                        parts++;
                    }
                if (parts > 1) tokens.push_back(BX_OR | parts);
                break;
            }
            case Node::kIntersect:
                for (const Node &k : n.kids) build(type, k);
                tokens.push_back(BX_AND | (uint32_t)n.kids.size());
                break;
            case Node::kExclude:
                build(type, n.kids[0]);
                build(type, n.kids[1]);
                tokens.push_back(BX_EXCL);
                break;
            case Node::kArrowAll: {
                // a.all(b): a leaf of its own whose only ops enumerate the tupleset, every child answering a result cell of ITS own
                // (OP_ALL); the leaf's value = NO if there is no child or one says NO, else ERR if one errs, else HAS (BX_LEAF_ALL)
                const Definition &def = sc.defs[type];
                const Member &rel = def.members[def.find(n.a)];
                const uint32_t saved = cur_leaf;
                cur_leaf = new_leaf();
                for (size_t k = 0; k < rel.classes.size(); k++) {
                    const int st = rel.classes[k].stype, tm = sc.defs[st].find(n.b);
                    if (tm < 0) continue;  // subject type lacks the computed permission: not dispatched (as for `->`; EXTERNAL, unverified)
                    row_op(OP_ENUM | OP_ALL, rel.slot, (int)k, 0, (uint32_t)sc.slot(st, tm));
                }
                tokens.push_back(BX_LEAF_ALL | cur_leaf);
                cur_leaf = saved;
                break;
            }
            default: break;  // (refs, arrows and nil are monotone)
        }
    }
    void root(int type, const Node &n) {
        combine = true;
        if (n.kind != Node::kUnion) {
            build(type, n);
            return;
        }
        uint32_t parts = 0;
        for (const Node &k : n.kids) {
            if (k.monotone()) expr(type, k, 0);  // direct
            else {
                build(type, k);
                parts++;
            }
        }
        if (parts > 1) tokens.push_back(BX_OR | parts);
    }
    void expr(int type, const Node &n, uint32_t d) {  // monotone sub-expressions only
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
                // a referenced permission with `&` / `-` of its own is never inlined: it is a state (and a combine node) of its own
                const bool nonmono = def.members[tm].is_permission && !def.members[tm].expr.monotone();
                if (recursive || nonmono || d + 1 > kMaxDepth || nops() >= kMaxOpsPerSlot) push_same(tslot, d);
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
            default: break;  // (kIntersect / kExclude never reach here: state() sends them to root(), refs to them are pushed)
        }
    }
};

void collect(const Node &n, Node::Kind kind, std::vector<const Node *> *out) {
    if (n.kind == kind || (kind == Node::kArrow && n.kind == Node::kArrowAll)) out->push_back(&n);  // (an intersection arrow walks its tupleset like any arrow)
    for (const Node &k : n.kids) collect(k, kind, out);
}

}  // namespace

// tables are sized for objects that do not exist yet, so that writes naming new objects can be patched in
// A quarter more than there is, and never fewer than 16 384 spare ids: the dual-write workflow names a new lock, a new workflow and two new
// activities per kube write (workflow.go:392-418, activity.go:81-102), so types that START empty grow by thousands of ids between two
// compactions -- with 1 024 spare ids the tables of `activity` were outgrown every ~500 kube writes, faster than a background build of the
// 10 M graph finishes (9 synchronous rebuilds in 37 k kube writes, profiles/r03_dual_write_before.json).  8 B per spare id and class.
uint32_t with_headroom(uint32_t n) { return n + n / 4 + 16384; }

uint32_t shard_of_type(const std::string &type_name, uint32_t world) {
    uint32_t h = 2166136261u;  // FNV-1a ...
    for (unsigned char c : type_name) h = (h ^ c) * 16777619u;
    // ... avalanched (the 32-bit murmur finaliser): raw FNV-1a of short names is poor in its low bits -- `pod`, `namespace`, `group` and `user` all
    // land on shard 0 of 2, of 3 and (but for `user`) of 4, so the small worlds the tests run never moved an entry between shards (found round 4)
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return world > 1 ? h % world : 0u;
}

void build_forward(Store &store, int64_t now, Snapshot *snap, ShardSpec shard) {
    store.settle_all();
    const Schema &sc = store.schema();
    Snapshot s;
    for (const Definition &d : sc.defs) s.type_owner.push_back(shard_of_type(d.name, shard.world));
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
    std::vector<uint32_t> cnt, fill;
    s.buckets.assign(4, kEmpty);  // bucket 0 is nobody's and stays empty: the kernels point a subject WITHOUT a row at {0, 1} and probe branch-free
    // pass A: which classes are live / hashed.  Sorted classes are numbered per TYPE: the row descriptors of all relations of
    // one object sit side by side ({start, end} pairs), so a state that touches two of them (`pod#view`: the group viewers and
    // the namespace arrow) finds both in one 16-byte record -- one cold line instead of two at level 1.
    std::vector<uint32_t> type_ks(sc.defs.size(), 0), type_base(sc.defs.size(), 0);
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        if (mem.is_permission) continue;
        RelLayout &l = lay[slot];
        l.nrows = with_headroom(store.objects(t).count());
        l.cls.resize(mem.classes.size());
        if (s.type_owner[t] != shard.rank) continue;  // another shard holds this type's rows
        for (const ClassTable &ct : tables[slot]) s.nedges_local += ct.keys.size();
        for (size_t k = 0; k < mem.classes.size(); k++) {
            ClassLayout &c = l.cls[k];
            const ClassTable &ct = tables[slot][k];
            c.hashed = mem.classes[k].srel == kNoRelation && !is_tupleset[slot];
            // Every class the schema DECLARES gets descriptors and program ops, relationships or not.  A class that
            // was skipped while empty made its first relationship unpatchable -- the parents' programs had no op for it -- and the next
            // fully-consistent read (check.go:41-46) rebuilt the whole snapshot: 100-140 ms on the 10 M graph.  The dual-write path hits
            // exactly that: `lock:<hash>#workflow@workflow:<id>` is created and deleted around every kube write (workflow.go:392-418), so on a
            // quiet proxy the class is empty at every build.  An empty class costs 8 B per object / subject of descriptors pointing at
            // nothing and one op per program that walks it.
            (void)ct;
            c.live = true;
            if (!c.hashed) c.ks = type_ks[t]++;
        }
    }
    for (size_t t = 0; t < sc.defs.size(); t++) {
        if (!type_ks[t]) continue;
        if (s.meta.size() % 4) s.meta.resize(s.meta.size() + 2, 0);  // 16-byte alignment: K == 2 rows load as one dwordx4
        type_base[t] = (uint32_t)(s.meta.size() / 2);
        s.meta.resize(s.meta.size() + 2 * (size_t)with_headroom(store.objects((int)t).count()) * type_ks[t], 0);
    }
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        if (mem.is_permission || s.type_owner[t] != shard.rank) continue;
        RelLayout &l = lay[slot];
        l.Ks = type_ks[t];
        l.meta_base = type_base[t];
        // ---- membership-only classes: one hashed row of RESOURCE ids per SUBJECT
        for (size_t k = 0; k < mem.classes.size(); k++) {
            ClassLayout &c = l.cls[k];
            if (!c.live || !c.hashed) continue;
            const ClassTable &ct = tables[slot][k];
            const bool filt = !ct.expiry.empty();
            // (a `T:*` class has ONE subject, the id of the name "*": its descriptor table ends there)
            const uint32_t ns = mem.classes[k].wildcard ? store.wildcard_id(mem.classes[k].stype) + 1 : with_headroom(store.objects(mem.classes[k].stype).count());
            c.nsubjects = ns;
            c.smeta_base = (uint32_t)(s.meta.size() / 2);
            s.meta.resize(s.meta.size() + 2 * (size_t)ns, 0);
            // group the class's resource ids by subject (counting sort), then lay the rows out one after the other
            cnt.assign((size_t)ns + 1, 0);
            for (uint64_t key : ct.keys)
                if ((uint32_t)key < ns && (!filt || store.live(ct, key, now))) cnt[(uint32_t)key + 1]++;
            for (uint32_t sid = 0; sid < ns; sid++) cnt[sid + 1] += cnt[sid];
            fill.assign(cnt.begin(), cnt.end() - 1);
            std::vector<uint32_t> by_subject(cnt[ns]);
            for (uint64_t key : ct.keys)
                if ((uint32_t)key < ns && (!filt || store.live(ct, key, now))) by_subject[fill[(uint32_t)key]++] = (uint32_t)(key >> 32);
            for (uint32_t sid = 0; sid < ns; sid++) {
                const auto r = append_hashed_row(s.buckets, by_subject.data() + cnt[sid], cnt[sid + 1] - cnt[sid], 0);
                uint32_t *md = s.meta.data() + 2 * ((size_t)c.smeta_base + sid);
                md[0] = r.first;
                md[1] = r.second;  // (y: buckets | two-choice | seed)
            }
        }
        // ---- enumerable classes: sorted sub-rows per (object, class)
        if (!l.Ks) continue;
        for (size_t k = 0; k < mem.classes.size(); k++) {
            const ClassLayout &c = l.cls[k];
            if (!c.live || c.hashed) continue;
            const ClassTable &ct = tables[slot][k];
            const bool filt = !ct.expiry.empty();
            uint32_t *meta = s.meta.data() + 2 * (size_t)l.meta_base;
            size_t i = 0;
            const size_t nk = ct.keys.size();
            while (i < nk) {  // one resource at a time (keys ascend by resource, then subject)
                const uint32_t res = (uint32_t)(ct.keys[i] >> 32);
                const uint32_t start = (uint32_t)s.edges.size();
                for (; i < nk && (uint32_t)(ct.keys[i] >> 32) == res; i++)
                    if (!filt || store.live(ct, ct.keys[i], now)) s.edges.push_back((uint32_t)ct.keys[i]);
                uint32_t *md = meta + 2 * ((size_t)res * l.Ks + c.ks);
                md[0] = start;
                md[1] = (uint32_t)s.edges.size();
            }
        }
    }
    s.nedges = 0;
    for (const auto &slot : tables)
        for (const auto &ct : slot) s.nedges += ct.keys.size();
    if (s.meta.empty()) s.meta.assign(4, 0);
    if (s.edges.empty()) s.edges.push_back(0);
    // ---- programs
    s.progs.resize(sc.nslots);
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        Flattener f(sc, lay, s.type_owner, shard.rank, &store);
        f.state(t, m, 0);
        SlotProg p{};
        p.owner = s.type_owner[t];
        p.first = (uint32_t)s.ops.size();
        p.n_probe = (uint32_t)f.probes.size();
        p.n_main = (uint32_t)(f.probes.size() + f.makers.size());
        p.n_total = (uint32_t)f.nops();
        p.max_dlevel = f.max_d;
        if (f.combine) {
            // A parent never evaluates a combine child's probes for it (they answer leaf cells that only exist once the state is visited):
            // no probe-only prefix, no leaf flags.  The boolean program: [ntokens][deepest offset per leaf 0..nleaves][tokens]
            p.n_probe = 0;
            p.nleaves = (uint32_t)f.leaf_maxd.size() - 1;
            if (s.bexpr.empty()) s.bexpr.push_back(0);
            p.combine = (uint32_t)s.bexpr.size();
            s.bexpr.push_back((uint32_t)f.tokens.size());
            s.bexpr.insert(s.bexpr.end(), f.leaf_maxd.begin(), f.leaf_maxd.end());
            s.bexpr.insert(s.bexpr.end(), f.tokens.begin(), f.tokens.end());
            s.has_combine = true;
        }
        s.ops.insert(s.ops.end(), f.probes.begin(), f.probes.end());
        s.ops.insert(s.ops.end(), f.makers.begin(), f.makers.end());
        s.ops.insert(s.ops.end(), f.reflex.begin(), f.reflex.end());
        s.progs[slot] = p;
    }
    if (s.ops.empty()) s.ops.push_back(FwdOp{});
    if (s.bexpr.empty()) s.bexpr.push_back(0);
    // which slots' values can depend on a combine program (their LookupResources is candidates + a forward Check): the combine slots and
    // whatever reaches one through a reference, an arrow or a userset subject
    s.slot_nonmono.assign(sc.nslots, 0);
    for (int slot = 0; slot < sc.nslots; slot++) s.slot_nonmono[slot] = s.progs[slot].combine ? 1 : 0;
    for (bool grew = s.has_combine; grew;) {
        grew = false;
        for (int slot = 0; slot < sc.nslots; slot++) {
            if (s.slot_nonmono[slot]) continue;
            auto [t, m] = sc.slot_owner[slot];
            const Member &mem = sc.defs[t].members[m];
            bool taint = false;
            if (!mem.is_permission) {
                for (const SubjectClass &c : mem.classes)
                    if (c.srel != kNoRelation && s.slot_nonmono[sc.slot(c.stype, c.srel)]) taint = true;
            } else {
                std::vector<const Node *> refs, arrows;
                collect(mem.expr, Node::kRef, &refs);
                collect(mem.expr, Node::kArrow, &arrows);
                for (const Node *r : refs)
                    if (s.slot_nonmono[sc.slot(t, sc.defs[t].find(r->a))]) taint = true;
                for (const Node *a : arrows)
                    for (const SubjectClass &c : sc.defs[t].members[sc.defs[t].find(a->a)].classes) {
                        const int tm = sc.defs[c.stype].find(a->b);
                        if (tm >= 0 && s.slot_nonmono[sc.slot(c.stype, tm)]) taint = true;
                    }
            }
            if (taint) {
                s.slot_nonmono[slot] = 1;
                grew = true;
            }
        }
    }
    // which slots' Checks can reach the depth limit: the longest chain of dependencies below a slot (every userset subject, reference and arrow counted as a
    // dispatch: an upper bound of what the walk counts), unbounded when the dependencies close a cycle
    {
        std::vector<std::vector<int>> deps(sc.nslots);
        for (int slot = 0; slot < sc.nslots; slot++) {
            auto [t, m] = sc.slot_owner[slot];
            const Member &mem = sc.defs[t].members[m];
            if (!mem.is_permission) {
                for (const SubjectClass &c : mem.classes)
                    if (c.srel != kNoRelation) deps[slot].push_back(sc.slot(c.stype, c.srel));
            } else {
                std::vector<const Node *> refs, arrows;
                collect(mem.expr, Node::kRef, &refs);
                collect(mem.expr, Node::kArrow, &arrows);
                for (const Node *r : refs) {
                    const int rm = sc.defs[t].find(r->a);
                    if (rm >= 0) deps[slot].push_back(sc.slot(t, rm));
                }
                for (const Node *a : arrows) {
                    const int am = sc.defs[t].find(a->a);
                    if (am < 0) continue;
                    deps[slot].push_back(sc.slot(t, am));
                    for (const SubjectClass &c : sc.defs[t].members[am].classes) {
                        const int tm = sc.defs[c.stype].find(a->b);
                        if (tm >= 0) deps[slot].push_back(sc.slot(c.stype, tm));
                    }
                }
            }
        }
        constexpr int kUnbounded = 1 << 20;
        std::vector<int> depth(sc.nslots, -1);  // -1 not visited, -2 on the stack
        std::function<int(int)> longest = [&](int v) -> int {
            if (depth[v] == -2) return kUnbounded;
            if (depth[v] >= 0) return depth[v];
            depth[v] = -2;
            int best = 0;
            for (int d : deps[v]) best = std::max(best, std::min(kUnbounded, longest(d) + 1));
            depth[v] = best;
            return best;
        };
        s.slot_deep.assign(sc.nslots, 0);
        for (int slot = 0; slot < sc.nslots; slot++) s.slot_deep[slot] = longest(slot) > 25 ? 1 : 0;
    }
    // ---- leaf bits: a userset edge `... @ T:c#m` always expands into the state (slot(T, m), c); mark the edges whose
    // child state has nothing left to enumerate, so the expansion can skip both the child's row descriptors and the
    // frontier write (kernels.hip, eval_child)
    auto row_nonempty = [&](const FwdOp &op, uint32_t id) {
        if (id >= op.nrows) return false;
        const uint32_t *md = s.meta.data() + 2 * ((size_t)op.base + (size_t)id * op.K + op.k);
        return md[1] > md[0];
    };
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        if (mem.is_permission) continue;
        const RelLayout &l = lay[slot];
        for (size_t k = 0; k < mem.classes.size(); k++) {
            const ClassLayout &c = l.cls[k];
            if (!c.live || c.hashed || mem.classes[k].srel == kNoRelation) continue;
            if (s.type_owner[mem.classes[k].stype] != shard.rank) continue;  // child rows are elsewhere: no leaf flags
            const SlotProg &tp = s.progs[sc.slot(mem.classes[k].stype, mem.classes[k].srel)];
            bool always = tp.combine != 0;  // (a combine child must be visited: its probes are its own business)
            std::vector<FwdOp> enums;
            for (uint32_t j = tp.n_probe; j < tp.n_main; j++) {
                const FwdOp &op = s.ops[tp.first + j];
                if (op.flags & OP_PUSH_SAME) always = true;
                else if (op.flags & OP_ENUM) enums.push_back(op);
            }
            if (always) continue;
            for (uint32_t id = 0; id < l.nrows; id++) {
                const uint32_t *md = s.meta.data() + 2 * ((size_t)l.meta_base + (size_t)id * l.Ks + c.ks);
                for (uint32_t e = md[0]; e < md[1]; e++) {
                    const uint32_t child = s.edges[e];
                    bool work = false;
                    for (const FwdOp &op : enums)
                        if (row_nonempty(op, child)) { work = true; break; }
                    if (!work) s.edges[e] = child | kLeafBit;
                }
            }
        }
    }
    if (getenv("ACL_DEBUG_ROWS")) {  // how the hashed rows were placed (tools/row_stats.py)
        uint64_t rows = 0, two = 0, ids = 0, nbs = 0, nb_two = 0;
        for (int slot = 0; slot < sc.nslots; slot++)
            for (const ClassLayout &c : lay[slot].cls) {
                if (!c.live || !c.hashed) continue;
                for (uint32_t sid = 0; sid < c.nsubjects; sid++) {
                    const uint32_t *md = s.meta.data() + 2 * ((size_t)c.smeta_base + sid);
                    if (!md[1]) continue;
                    rows++;
                    nbs += hrow_nb(md[1]);
                    if (md[1] & kRowTwoBit) two++, nb_two += hrow_nb(md[1]);
                    for (size_t i = 4 * (size_t)md[0]; i < 4 * ((size_t)md[0] + hrow_nb(md[1])); i++) ids += s.buckets[i] != kEmpty;
                }
            }
        fprintf(stderr, "[aclgpu] hashed rows: %llu (%llu two-choice holding %llu buckets), %llu ids in %llu buckets (load %.3f), %.1f MB\n", (unsigned long long)rows,
                (unsigned long long)two, (unsigned long long)nb_two, (unsigned long long)ids, (unsigned long long)nbs, nbs ? ids / (4.0 * nbs) : 0.0, nbs * 16 / 1e6);
    }
    s.lay = std::move(lay);
    *snap = std::move(s);
}


// ---------------------------------------------------------------------------------------------- patching
namespace {

constexpr size_t kMaxPatchChanges = 8192;  // beyond this a rebuild is cheaper than row-by-row patching

struct Patcher {
    Store &store;
    Snapshot &s;
    std::vector<Patch> &out;
    bool ops_dirty = false;

    // ---- hashed rows (subject-indexed): row of `sid` holds resource ids
    uint32_t *hdesc(const ClassLayout &c, uint32_t sid) { return s.meta.data() + 2 * ((size_t)c.smeta_base + sid); }
    bool hashed_has(const ClassLayout &c, uint32_t sid, uint32_t res) {
        const uint32_t *md = hdesc(c, sid);
        return hashed_row_has(s.buckets.data() + 4 * (size_t)md[0], md[1], res);
    }
    // rebuilds the row of `sid` with `res` added or removed: in place while it fits its buckets (a seeded row gets a new seed when the old
    // one no longer places every id: the descriptor is re-uploaded with the row), else at the end of `buckets` with room to grow before the next move
    void hashed_set(const ClassLayout &c, uint32_t sid, uint32_t res, bool add) {
        uint32_t *md = hdesc(c, sid);
        const uint32_t b0 = md[0], nb = hrow_nb(md[1]);
        std::vector<uint32_t> el;
        for (size_t i = 4 * (size_t)b0; i < 4 * (size_t)(b0 + nb); i++)
            if (s.buckets[i] != kEmpty && (add || s.buckets[i] != res)) el.push_back(s.buckets[i]);
        if (add) el.push_back(res);
        if (nb) {
            static thread_local std::vector<uint8_t> fill;
            std::vector<uint32_t> row(4 * (size_t)nb);
            const uint32_t y = place_row(row.data(), nb, el.data(), el.size(), seeded_rows(), fill);
            if (y) {
                std::copy(row.begin(), row.end(), s.buckets.begin() + 4 * (long)b0);
                out.push_back(Patch{Patch::BUCKETS, 4 * (size_t)b0, 4 * (size_t)nb});
                if (y != md[1]) {
                    md[1] = y;
                    out.push_back(Patch{Patch::META, (size_t)(md - s.meta.data()), 2});
                }
                return;
            }
        }
        const auto r = append_hashed_row(s.buckets, el.data(), el.size(), buckets_for((uint32_t)(el.size() + el.size() / 2 + 1)));
        md = hdesc(c, sid);
        md[0] = r.first;
        md[1] = r.second;
        s.garbage_words += 4 * (uint64_t)nb;
        out.push_back(Patch{Patch::BUCKETS, 4 * (size_t)r.first, 4 * (size_t)hrow_nb(r.second)});
        out.push_back(Patch{Patch::META, (size_t)(md - s.meta.data()), 2});
    }

    // ---- sorted rows: row of (res, class) holds subject ids ascending (| leaf flag)
    uint32_t *sdesc(const RelLayout &l, const ClassLayout &c, uint32_t res) { return s.meta.data() + 2 * ((size_t)l.meta_base + (size_t)res * l.Ks + c.ks); }
    bool sorted_find(const uint32_t *md, uint32_t sid, uint32_t *pos) {
        uint32_t lo = md[0], hi = md[1];
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if ((s.edges[mid] & kIdMask) < sid) lo = mid + 1;
            else hi = mid;
        }
        *pos = lo;
        return lo < md[1] && (s.edges[lo] & kIdMask) == sid;
    }
    bool row_nonempty(const FwdOp &op, uint32_t id) const {
        if (id >= op.nrows) return false;
        const uint32_t *md = s.meta.data() + 2 * ((size_t)op.base + (size_t)id * op.K + op.k);
        return md[1] > md[0];
    }
    // leaf flag of a new userset edge -> child (target slot, child id): nothing left to enumerate there
    bool child_is_leaf(uint32_t target_slot, uint32_t child) const {
        const SlotProg &tp = s.progs[target_slot];
        if (tp.combine) return false;
        for (uint32_t j = tp.n_probe; j < tp.n_main; j++) {
            const FwdOp &op = s.ops[tp.first + j];
            if (op.flags & OP_PUSH_SAME) return false;
            if ((op.flags & OP_ENUM) && row_nonempty(op, child)) return false;
        }
        return true;
    }
    // an object of type t got its first enumerable row: edges elsewhere may still flag it as a leaf.  Leaf flags are
    // an optimisation; switch their authority off for every op whose children are of type t (the kernel then looks
    // at the child's rows, which are exact) until the next rebuild recomputes them.
    void distrust_leaf_flags(int t) {
        const Schema &sc = store.schema();
        for (FwdOp &op : s.ops)
            if ((op.flags & OP_ENUM) && (op.flags & OP_LEAFBIT) && sc.slot_owner[op.key].first == t) {
                op.flags &= ~(uint32_t)OP_LEAFBIT;
                ops_dirty = true;
            }
    }
    void sorted_add(int slot, int cls, const RelLayout &l, const ClassLayout &c, uint32_t res, uint32_t sid) {
        const Schema &sc = store.schema();
        auto [t, m] = sc.slot_owner[slot];
        const SubjectClass &k = sc.defs[t].members[m].classes[cls];
        uint32_t *md = sdesc(l, c, res);
        uint32_t pos;
        sorted_find(md, sid, &pos);
        const uint32_t a = md[0], b = md[1];
        uint32_t edge = sid;
        if (k.srel != kNoRelation && s.type_owner[k.stype] == s.type_owner[t] && child_is_leaf((uint32_t)sc.slot(k.stype, k.srel), sid)) edge |= kLeafBit;
        const uint32_t len = b - a, ipos = pos - a;
        auto capit = s.edge_cap.find(a);
        const uint32_t cap = capit == s.edge_cap.end() ? len : capit->second;
        if (len && len + 1 <= cap) {  // the row was moved before and has room: insert in place
            std::copy_backward(s.edges.begin() + pos, s.edges.begin() + b, s.edges.begin() + b + 1);
            s.edges[pos] = edge;
            md[1] = b + 1;
            out.push_back(Patch{Patch::EDGES, pos, (size_t)(b + 1 - pos)});
            out.push_back(Patch{Patch::META, (size_t)(md - s.meta.data()), 2});
        } else {  // move it to the end of `edges` with the subject inserted and room for half as many again
            const uint32_t start = (uint32_t)s.edges.size(), ncap = std::max<uint32_t>(4, (len + 1) + (len + 1) / 2);
            std::vector<uint32_t> row(s.edges.begin() + a, s.edges.begin() + b);  // (copy first: the append may reallocate)
            row.insert(row.begin() + ipos, edge);
            s.edges.insert(s.edges.end(), row.begin(), row.end());
            s.edges.resize((size_t)start + ncap, 0u);
            if (capit != s.edge_cap.end()) s.edge_cap.erase(capit);
            s.edge_cap[start] = ncap;
            md = sdesc(l, c, res);
            md[0] = start;
            md[1] = start + len + 1;
            s.garbage_words += cap;
            out.push_back(Patch{Patch::EDGES, start, (size_t)(len + 1)});
            out.push_back(Patch{Patch::META, (size_t)(md - s.meta.data()), 2});
        }
        if (a == b) distrust_leaf_flags(t);  // `res` may just have stopped being a leaf
    }
    void sorted_remove(const RelLayout &l, const ClassLayout &c, uint32_t res, uint32_t sid) {
        uint32_t *md = sdesc(l, c, res);
        uint32_t pos;
        if (!sorted_find(md, sid, &pos)) return;
        std::copy(s.edges.begin() + pos + 1, s.edges.begin() + md[1], s.edges.begin() + pos);
        md[1]--;
        s.garbage_words++;
        if (md[1] > md[0]) out.push_back(Patch{Patch::EDGES, md[0], (size_t)(md[1] - md[0])});
        out.push_back(Patch{Patch::META, (size_t)(md - s.meta.data()), 2});
        // a row that became empty leaves stale NON-leaf flags behind: harmless (one wasted frontier entry)
    }
};

}  // namespace

void expiry_crossings(const Store &store, int64_t lo, int64_t hi, int64_t now, std::vector<Store::Change> *ch) { store.expiry_crossings(lo, hi, now, ch); }

// ACL_DEBUG_REBUILD: say why a snapshot could not be patched (the caller then rebuilds, or drops a background build)
static bool unpatchable(const char *why, uint64_t a = 0, uint64_t b = 0) {
    static const bool on = getenv("ACL_DEBUG_REBUILD") != nullptr;
    if (on) fprintf(stderr, "[aclgpu] patch_forward declines: %s (%llu, %llu)\n", why, (unsigned long long)a, (unsigned long long)b);
    return false;
}

bool patch_forward(Store &store, int64_t now, Snapshot *snap, ShardSpec shard, std::vector<Patch> *patches, size_t max_changes) {
    Snapshot &s = *snap;
    std::vector<Store::Change> ch;
    if (s.lay.empty()) return unpatchable("no layout");
    if (!store.raw_changes_since(s.revision, &ch)) return unpatchable("change feed does not reach back to the snapshot's revision (bulk load / dropped window)", s.revision, store.revision());
    expiry_crossings(store, s.valid_lo, s.valid_hi, now, &ch);
    if (ch.size() > (max_changes ? max_changes : kMaxPatchChanges)) return unpatchable("too many changes", ch.size(), max_changes ? max_changes : kMaxPatchChanges);
    store.settle_all();
    const Schema &sc = store.schema();
    // objects created since the build must fit the headroom of every table they index
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        if (mem.is_permission || s.type_owner[t] != shard.rank) continue;
        const RelLayout &l = s.lay[slot];
        if (store.objects(t).count() > l.nrows) return unpatchable("objects of a relation's type outgrew its row table", store.objects(t).count(), l.nrows);
        for (size_t k = 0; k < mem.classes.size(); k++)
            if (l.cls[k].live && l.cls[k].hashed && !mem.classes[k].wildcard && store.objects(mem.classes[k].stype).count() > l.cls[k].nsubjects)
                return unpatchable("subjects of a hashed class outgrew its descriptor table", store.objects(mem.classes[k].stype).count(), l.cls[k].nsubjects);
    }
    std::sort(ch.begin(), ch.end(), [](const Store::Change &a, const Store::Change &b) {
        return a.slot != b.slot ? a.slot < b.slot : a.cls != b.cls ? a.cls < b.cls : a.key < b.key;
    });
    auto &tables = store.tables();
    Patcher P{store, s, *patches};
    // first pass: can everything be expressed as a patch?  (nothing is modified before we know)
    for (const Store::Change &c : ch) {
        const int t = sc.slot_owner[c.slot].first;
        if (s.type_owner[t] != shard.rank) continue;
        const ClassTable &ct = tables[c.slot][c.cls];
        if (!s.lay[c.slot].cls[c.cls].live && ct.contains(c.key) && store.live(ct, c.key, now)) return unpatchable("relationship in a class without rows");  // (cannot happen for a class of an owned type: all are live)
    }
    for (size_t i = 0; i < ch.size(); i++) {
        const Store::Change &c = ch[i];
        if (i && ch[i - 1].slot == c.slot && ch[i - 1].cls == c.cls && ch[i - 1].key == c.key) continue;  // net effect per relationship
        const int t = sc.slot_owner[c.slot].first;
        if (s.type_owner[t] != shard.rank) continue;
        const RelLayout &l = s.lay[c.slot];
        const ClassLayout &cl = l.cls[c.cls];
        if (!cl.live) continue;  // deletions in a class that was already empty
        const ClassTable &ct = tables[c.slot][c.cls];
        const bool want = ct.contains(c.key) && store.live(ct, c.key, now);
        const uint32_t res = (uint32_t)(c.key >> 32), sid = (uint32_t)c.key;
        if (cl.hashed) {
            if (want != P.hashed_has(cl, sid, res)) {
                P.hashed_set(cl, sid, res, want);
                s.patched++;
            }
        } else {
            uint32_t pos;
            const bool have = P.sorted_find(P.sdesc(l, cl, res), sid, &pos);
            if (want && !have) P.sorted_add(c.slot, c.cls, l, cl, res, sid);
            else if (!want && have) P.sorted_remove(l, cl, res, sid);
            if (want != have) s.patched++;
        }
    }
    if (P.ops_dirty) patches->push_back(Patch{Patch::OPS, 0, s.ops.size()});
    s.patch_lo = s.valid_lo;
    s.patch_hi = s.valid_hi;
    store.expiry_window(now, &s.valid_lo, &s.valid_hi);
    s.revision = store.revision();
    s.nedges = s.nedges_local = 0;
    for (int slot = 0; slot < sc.nslots; slot++)
        for (const ClassTable &ct : tables[slot]) {
            s.nedges += ct.keys.size();
            if (s.type_owner[sc.slot_owner[slot].first] == shard.rank) s.nedges_local += ct.keys.size();
        }
    for (size_t t = 0; t < sc.defs.size(); t++) s.type_nobjects[t] = store.objects((int)t).count();
    return true;  // the reverse rows (if built) are now one feed position behind: patch_reverse or rebuild them
}

// Test hook (acl_selfcheck_snapshot): does the snapshot -- however it got here, built or patched -- hold exactly the
// store's live relationships, in rows the kernels can search, with only sound leaf flags?
bool verify_snapshot(Store &store, int64_t now, const Snapshot &s, ShardSpec shard, std::string *why) {
    store.settle_all();
    const Schema &sc = store.schema();
    auto &tables = store.tables();
    auto bad = [&](const std::string &m) {
        if (why) *why = m;
        return false;
    };
    if (s.lay.size() != (size_t)sc.nslots) return bad("layout does not match the schema");
    Snapshot &ms = const_cast<Snapshot &>(s);
    std::vector<Patch> none;
    Patcher P{store, ms, none};
    for (int slot = 0; slot < sc.nslots; slot++) {
        auto [t, m] = sc.slot_owner[slot];
        const Member &mem = sc.defs[t].members[m];
        if (mem.is_permission || s.type_owner[t] != shard.rank) continue;
        const RelLayout &l = s.lay[slot];
        const std::string rel = sc.defs[t].name + "#" + mem.name;
        for (size_t k = 0; k < mem.classes.size(); k++) {
            const ClassLayout &c = l.cls[k];
            const ClassTable &ct = tables[slot][k];
            std::vector<uint64_t> live;
            for (uint64_t key : ct.keys)
                if (store.live(ct, key, now)) live.push_back(key);
            if (!c.live) {
                if (!live.empty()) return bad(rel + ": class has relationships but no rows");
                continue;
            }
            size_t stored = 0;
            if (c.hashed) {
                for (uint64_t key : live)
                    if ((uint32_t)key >= c.nsubjects || !P.hashed_has(c, (uint32_t)key, (uint32_t)(key >> 32))) return bad(rel + ": hashed row misses a relationship");
                for (uint32_t sid = 0; sid < c.nsubjects; sid++) {
                    const uint32_t *md = s.meta.data() + 2 * ((size_t)c.smeta_base + sid);
                    if (4 * ((size_t)md[0] + hrow_nb(md[1])) > s.buckets.size()) return bad(rel + ": hashed descriptor out of range");
                    for (size_t i = 4 * (size_t)md[0]; i < 4 * ((size_t)md[0] + hrow_nb(md[1])); i++)
                        if (s.buckets[i] != 0xFFFFFFFFu) {
                            stored++;
                            if (!std::binary_search(live.begin(), live.end(), (uint64_t)s.buckets[i] << 32 | sid)) return bad(rel + ": hashed row holds a dead relationship");
                        }
                }
            } else {
                for (uint32_t res = 0; res < l.nrows; res++) {
                    const uint32_t *md = s.meta.data() + 2 * ((size_t)l.meta_base + (size_t)res * l.Ks + c.ks);
                    if (md[1] < md[0] || md[1] > s.edges.size()) return bad(rel + ": row descriptor out of range");
                    for (uint32_t e = md[0]; e < md[1]; e++) {
                        const uint32_t id = s.edges[e] & kIdMask;
                        if (e > md[0] && (s.edges[e - 1] & kIdMask) >= id) return bad(rel + ": row not strictly ascending");
                        if (!std::binary_search(live.begin(), live.end(), (uint64_t)res << 32 | id)) return bad(rel + ": row holds a dead relationship");
                        stored++;
                    }
                }
            }
            if (stored != live.size()) return bad(rel + ": " + std::to_string(stored) + " stored vs " + std::to_string(live.size()) + " live relationships");
        }
    }
    // reverse rows (when built): subject -> ascending resource ids, exactly the live relationships
    if (s.has_reverse) {
        if (s.rlay.size() != (size_t)sc.nslots) return bad("reverse layout does not match the schema");
        for (int slot = 0; slot < sc.nslots; slot++) {
            auto [t, m] = sc.slot_owner[slot];
            const Member &mem = sc.defs[t].members[m];
            if (mem.is_permission) continue;
            const std::string rel = sc.defs[t].name + "#" + mem.name + " (reverse)";
            if (store.objects(t).count() > s.slot_nobjects[slot]) return bad(rel + ": visited bitmap too small for the type's objects");
            if (s.type_owner[t] != shard.rank) continue;
            for (size_t k = 0; k < mem.classes.size(); k++) {
                const Snapshot::RevLayout &l = s.rlay[slot][k];
                const ClassTable &ct = tables[slot][k];
                size_t live = 0, stored = 0;
                for (uint64_t key : ct.keys) live += store.live(ct, key, now) ? 1 : 0;
                if (!l.any) {
                    if (live) return bad(rel + ": class has relationships but no reverse rows");
                    continue;
                }
                for (uint32_t sid = 0; sid < l.nrows; sid++) {
                    const uint32_t *md = s.rmeta.data() + 2 * ((size_t)l.base + sid);
                    if (md[1] < md[0] || md[1] > s.redges.size()) return bad(rel + ": descriptor out of range");
                    for (uint32_t e = md[0]; e < md[1]; e++) {
                        if (e > md[0] && s.redges[e - 1] >= s.redges[e]) return bad(rel + ": row not strictly ascending");
                        const uint64_t key = (uint64_t)s.redges[e] << 32 | sid;
                        if (!ct.contains(key) || !store.live(ct, key, now)) return bad(rel + ": row holds a dead relationship");
                        stored++;
                    }
                }
                if (stored != live) return bad(rel + ": " + std::to_string(stored) + " stored vs " + std::to_string(live) + " live relationships");
            }
        }
    }
    // leaf flags: wherever an op still trusts them, a flagged child must really have nothing to enumerate
    for (const FwdOp &op : s.ops) {
        if (!(op.flags & OP_ENUM) || !(op.flags & OP_LEAFBIT)) continue;
        for (uint32_t res = 0; res < op.nrows; res++) {
            const uint32_t *md = s.meta.data() + 2 * ((size_t)op.base + (size_t)res * op.K + op.k);
            for (uint32_t e = md[0]; e < md[1]; e++)
                if ((s.edges[e] & kLeafBit) && !P.child_is_leaf(op.key, s.edges[e] & kIdMask)) return bad("stale leaf flag under an op that trusts leaf flags");
        }
    }
    return true;
}

}  // namespace acl
