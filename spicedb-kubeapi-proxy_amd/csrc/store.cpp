// store.cpp -- see store.hpp.  Error codes follow what the reference's callers
// observe from SpiceDB at the seam (SURVEY.md 8(b) "Error conventions").
#include "store.hpp"

#include "validate.hpp"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstring>
#include <cstdlib>
#include <new>
#include <sys/mman.h>

#include "../../include/aclgpu.h"

namespace acl {

// ---------------------------------------------------------------- ObjectTable
// 8 bytes per step through a 64 x 64 -> 128-bit multiply-fold (the construction of wyhash / rapidhash): a 15-byte object id is two steps and
// a finaliser, where byte-at-a-time FNV-1a was a chain of 15 dependent multiplies -- two ids per item made that half of the string entry
// point's host time.  (The function is internal to the table: nothing outside depends on its values.)
static inline uint64_t mulfold(uint64_t a, uint64_t b) {
    const __uint128_t r = (__uint128_t)a * b;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}
uint64_t ObjectTable::hash(std::string_view s) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(s.data());
    size_t n = s.size();
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
    while (n >= 8) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        h = mulfold(h ^ v, 0xE7037ED1A0B428DBull);
        p += 8;
        n -= 8;
    }
    if (n) {
        uint64_t v = 0;
        std::memcpy(&v, p, n);  // (little-endian tail, zero-padded: the length is mixed in above)
        h = mulfold(h ^ v, 0x8EBC6AF09C88C6E3ull);
    }
    return mulfold(h, 0x589965CC75374CC3ull) ^ h;
}
void *slot_pages_alloc(size_t bytes) {
    constexpr size_t kHuge = (size_t)2 << 20;
    if (bytes < kHuge) {
        void *p = nullptr;
        if (posix_memalign(&p, 64, bytes) != 0) throw std::bad_alloc();
        return p;
    }
    const size_t len = (bytes + kHuge - 1) & ~(kHuge - 1);
    // (over-map by one huge page and trim both ends: mmap only promises 4 KiB alignment)
    char *raw = static_cast<char *>(mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (raw == MAP_FAILED) throw std::bad_alloc();
    char *p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
    if (p != raw) munmap(raw, (size_t)(p - raw));
    if (p + len != raw + len + kHuge) munmap(p + len, (size_t)(raw + len + kHuge - (p + len)));
    static const bool kAsk = [] {
        const char *e = getenv("ACL_SLOT_HUGE_PAGES");  // (A/B knob: 0 leaves the table on 4 KiB pages)
        return !(e && atoi(e) == 0);
    }();
    if (kAsk) (void)madvise(p, len, MADV_HUGEPAGE);  // (advice only: refused on kernels without THP)
    return p;
}
void slot_pages_free(void *p, size_t bytes) {
    constexpr size_t kHuge = (size_t)2 << 20;
    if (bytes < kHuge) std::free(p);
    else munmap(p, (bytes + kHuge - 1) & ~(kHuge - 1));
}
void ObjectTable::grow() {
    std::vector<Slot, SlotPages<Slot>> old;
    old.swap(slots_);
    // (doubling while small; a quarter more, rounded to whole 4 KiB pages of slots, from kBigTable slots on -- see store.hpp)
    const size_t cap = old.empty() ? 64 : (old.size() * 2 <= kBigTable ? old.size() * 2 : ((old.size() + old.size() / 4 + 63) / 64) * 64);
    slots_.assign(cap, Slot{0, 0xFFFFFFFFu, 0, {}, nullptr});
    used_ -= tombs_;
    tombs_ = 0;
    for (const Slot &s : old) {
        if (s.id == 0xFFFFFFFFu || s.id == kTomb) continue;
        size_t i = home(hash(names_[name_of_[s.id]]));
        while (slots_[i].id != 0xFFFFFFFFu) i = next(i);
        slots_[i] = s;
    }
}
bool ObjectTable::find(std::string_view name, uint32_t *id) const { return find_hashed(name, hash(name), id); }
bool ObjectTable::find_hashed(std::string_view name, uint64_t h, uint32_t *id) const {
    if (slots_.empty()) return false;
    const uint32_t tag = (uint32_t)(h >> 32);
    for (size_t i = home(h);; i = next(i)) {
        const Slot &s = slots_[i];
        if (s.id == 0xFFFFFFFFu) return false;
        if (s.tag == tag && s.id != kTomb) {
            const size_t n = name.size();
            if (n <= kInline) {
                if (s.len == n && !std::memcmp(s.inl, name.data(), n)) {
                    *id = s.id;
                    return true;
                }
            } else if (s.len == std::min<size_t>(n, 0xFFFFu) && !std::memcmp(s.inl, name.data(), kInline)) {
                // (the stored name is NUL-terminated; never read past that NUL: a longer query must not run off a shorter name)
                const char *a = s.far, *b = name.data();
                size_t k = kInline;
                while (k < n && a[k] == b[k] && a[k] != 0) k++;
                if (k == n && a[n] == 0) {
                    *id = s.id;
                    return true;
                }
            }
        }
    }
}
ObjectTable::Slot ObjectTable::make_slot(uint64_t h, uint32_t id, const std::string &stored) {
    Slot s{(uint32_t)(h >> 32), id, (uint16_t)std::min<size_t>(stored.size(), 0xFFFFu), {}, nullptr};
    std::memcpy(s.inl, stored.data(), std::min<size_t>(stored.size(), kInline));
    if (stored.size() > kInline) s.far = stored.c_str();
    return s;
}
uint32_t ObjectTable::intern(std::string_view name) {
    uint32_t id;
    if (find(name, &id)) return id;
    if (slots_.empty() || full_for_one_more()) grow();
    id = count();
    if (name_of_.size() <= id) name_of_.resize((size_t)id + 1, 0xFFFFFFFFu);
    name_of_[id] = (uint32_t)names_.size();
    names_.emplace_back(name);
    const uint64_t h = hash(name);
    size_t i = home(h);
    while (slots_[i].id != 0xFFFFFFFFu && slots_[i].id != kTomb) i = next(i);
    if (slots_[i].id == kTomb) tombs_--;
    else used_++;
    slots_[i] = make_slot(h, id, names_.back());
    count_.store(id + 1, std::memory_order_release);
    return id;
}
void ObjectTable::rename(uint32_t id, std::string_view new_name) {
    std::string &stored = names_[name_of_[id]];
    {   // the old name's slot becomes a tombstone
        const uint64_t h = hash(stored);
        for (size_t i = home(h);; i = next(i)) {
            Slot &s = slots_[i];
            if (s.id == 0xFFFFFFFFu) break;  // (cannot happen: the name is in the table)
            if (s.id == id) {
                s = Slot{0, kTomb, 0, {}, nullptr};
                tombs_++;
                break;
            }
        }
    }
    stored.assign(new_name.data(), new_name.size());
    if (tombs_ * 4 > slots_.size()) {  // tombstones lengthen every probe: re-hash in place
        std::vector<Slot, SlotPages<Slot>> old;
        old.swap(slots_);
        slots_.assign(old.size(), Slot{0, 0xFFFFFFFFu, 0, {}, nullptr});
        used_ -= tombs_;
        tombs_ = 0;
        for (const Slot &s : old) {
            if (s.id == 0xFFFFFFFFu || s.id == kTomb) continue;
            size_t i = home(hash(names_[name_of_[s.id]]));
            while (slots_[i].id != 0xFFFFFFFFu) i = next(i);
            slots_[i] = s;
        }
    }
    if (full_for_one_more()) grow();
    const uint64_t h = hash(new_name);
    size_t i = home(h);
    while (slots_[i].id != 0xFFFFFFFFu && slots_[i].id != kTomb) i = next(i);
    if (slots_[i].id == kTomb) tombs_--;
    else used_++;
    slots_[i] = make_slot(h, id, stored);
}
const std::string *ObjectTable::name(uint32_t id) const {
    if (id >= name_of_.size() || name_of_[id] == 0xFFFFFFFFu) return nullptr;
    return &names_[name_of_[id]];
}

// ----------------------------------------------------------------- ClassTable
void ClassTable::settle() {
    if (pending.empty()) return;
    std::sort(pending.begin(), pending.end());
    pending.erase(std::unique(pending.begin(), pending.end()), pending.end());
    if (keys.empty()) {
        keys.mut().swap(pending);
    } else {
        std::vector<uint64_t> merged;
        merged.reserve(keys.size() + pending.size());
        std::set_union(keys.begin(), keys.end(), pending.begin(), pending.end(), std::back_inserter(merged));
        keys.mut().swap(merged);
    }
    pending.clear();
    pending.shrink_to_fit();
}
bool ClassTable::contains(uint64_t k) const { return std::binary_search(keys.begin(), keys.end(), k); }

// ---------------------------------------------------------------------- text
bool parse_relationship_text(const std::string &line, RelText *out) {
    // grammar of pkg/rules/rules.go:1053-1055 (lazy groups => split at the FIRST separator)
    size_t c1 = line.find(':');
    if (c1 == std::string::npos) return false;
    size_t h1 = line.find('#', c1 + 1);
    if (h1 == std::string::npos) return false;
    size_t at = line.find('@', h1 + 1);
    if (at == std::string::npos) return false;
    size_t c2 = line.find(':', at + 1);
    if (c2 == std::string::npos) return false;
    size_t h2 = line.find('#', c2 + 1);
    out->rtype = line.substr(0, c1);
    out->rid = line.substr(c1 + 1, h1 - c1 - 1);
    out->rel = line.substr(h1 + 1, at - h1 - 1);
    out->stype = line.substr(at + 1, c2 - at - 1);
    if (h2 == std::string::npos) {
        out->sid = line.substr(c2 + 1);
        out->srel.clear();
    } else {
        out->sid = line.substr(c2 + 1, h2 - c2 - 1);
        out->srel = line.substr(h2 + 1);
    }
    out->expires_at = 0;
    return true;
}

// ---------------------------------------------------------------------- Store
Status Store::load_schema(const std::string &text) {
    Schema s;
    std::string err;
    if (!parse_schema(text, &s, &err)) return Status::Err(ACL_ERR_INVALID_ARGUMENT, err);
    schema_ = std::move(s);
    schema_loaded_ = true;
    objects_.assign(schema_.defs.size(), ObjectTable());
    tables_.assign(schema_.nslots, {});
    expiry_index_.clear();
    for (int slot = 0; slot < schema_.nslots; slot++) {
        auto [t, m] = schema_.slot_owner[slot];
        tables_[slot].assign(schema_.defs[t].members[m].classes.size(), ClassTable());
    }
    // `T:*` subjects: the name "*" gets an id in T's table right away, so that programs built before the first wildcard relationship
    // exists can already name the wildcard subject's row
    refcnt_.assign(schema_.defs.size(), {});
    freed_.assign(schema_.defs.size(), {});
    freed_at_.assign(schema_.defs.size(), {});
    touched_at_.assign(schema_.defs.size(), {});
    no_recycle_.assign(schema_.defs.size(), 0);
    recycled_rev_.clear();
    if (const char *ev = getenv("ACL_ID_QUARANTINE_MS")) reuse_quarantine_ms_ = std::max(0, atoi(ev));  // test knob
    wildcard_id_.assign(schema_.defs.size(), 0xFFFFFFFFu);
    for (const Definition &d : schema_.defs)
        for (const Member &m : d.members)
            for (const SubjectClass &c : m.classes)
                if (c.wildcard && wildcard_id_[c.stype] == 0xFFFFFFFFu) wildcard_id_[c.stype] = intern_object(c.stype, "*", true);
    tupleset_slot_.assign(schema_.nslots, 0);
    for (size_t t = 0; t < schema_.defs.size(); t++)
        for (const Member &m : schema_.defs[t].members) {
            if (!m.is_permission) continue;
            std::function<void(const Node &)> arrows = [&](const Node &n) {
                if (n.kind == Node::kArrow || n.kind == Node::kArrowAll) {
                    const int am = schema_.defs[t].find(n.a);
                    if (am >= 0) tupleset_slot_[schema_.defs[t].members[am].slot] = 1;
                }
                for (const Node &k : n.kids) arrows(k);
            };
            arrows(m.expr);
        }
    path_adds_++;
    revision_++;
    log_.clear();  // ids of the previous schema mean nothing now
    log_floor_ = revision_;
    return Status::Ok();
}

int64_t Store::steady_now_ms() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int64_t steady_ms() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void Store::ref(int type, uint32_t id, int delta) {
    if (no_recycle_[type]) return;
    auto &rc = refcnt_[type];
    if (rc.size() <= id) rc.resize((size_t)id + 1 + rc.size() / 2, 0);
    uint32_t &c = rc[id];
    if (delta > 0) {
        c++;
    } else if ((c & ~kPinned) > 0) {
        c--;
        if (c == 0) note_free(type, id);  // (pinned ids never reach 0)
    }
}
// the object just became (or was born) unreferenced: on the free list, stamped -- an older entry of the same id is void from here on
void Store::note_free(int type, uint32_t id) {
    auto &fa = freed_at_[type];
    if (fa.size() <= id) fa.resize((size_t)id + 1 + fa.size() / 2, 0);
    const int64_t t = steady_ms();
    fa[id] = t;
    auto &fq = freed_[type];
    fq.push_back(Freed{id, t});
    // Entries go void when their object is referenced again (or freed again: the younger entry counts) and are only dropped when a NEW name
    // looks for an id -- a fixed set of objects whose relationships come and go for ever would grow the list by an entry per round.  Past
    // twice the type's id space the void ones are swept out.
    if (fq.size() > 2 * (size_t)objects_[type].count() + 4096) {
        std::deque<Freed> keep;
        const auto &rc = refcnt_[type];
        for (const Freed &f : fq)
            if (f.id < rc.size() && rc[f.id] == 0 && f.id < fa.size() && fa[f.id] == f.at_ms && (keep.empty() || keep.back().id != f.id || keep.back().at_ms != f.at_ms)) keep.push_back(f);
        fq.swap(keep);
    }
}
void Store::ref_key(int slot, int cls, uint64_t key, int delta) {
    auto [t, m] = schema_.slot_owner[slot];
    ref(t, (uint32_t)(key >> 32), delta);
    ref(schema_.defs[t].members[m].classes[cls].stype, (uint32_t)key, delta);
}

uint32_t Store::intern_object(int type, std::string_view name, bool pin, bool hold) {
    ObjectTable &tab = objects_[type];
    uint32_t id;
    if (tab.find(name, &id)) {
        if (pin && !no_recycle_[type]) {
            auto &rc = refcnt_[type];
            if (rc.size() <= id) rc.resize((size_t)id + 1 + rc.size() / 2, 0);
            rc[id] |= kPinned;
        }
        if (hold) ref(type, id, +1);  // (every hold is released once: a write naming one new subject in five updates holds it five times)
        touch(type, id);  // (found by name: the quarantine of an unreferenced object starts over)
        return id;
    }
    // a NEW name: the oldest free id that has sat out its quarantine, if its object is still unreferenced; else the next dense id
    auto &fq = freed_[type];
    const int64_t now_ms = steady_ms();
    while (!no_recycle_[type] && !fq.empty() && now_ms - fq.front().at_ms >= reuse_quarantine_ms_) {
        const Freed f = fq.front();
        fq.pop_front();
        // void entries: referenced (or pinned) again since, or freed AGAIN later (the younger entry carries the quarantine)
        if (f.id >= refcnt_[type].size() || refcnt_[type][f.id] != 0 || !tab.name(f.id) || f.id >= freed_at_[type].size() || freed_at_[type][f.id] != f.at_ms) continue;
        // handed out by name since it became free (touch): the quarantine counts from then -- back into the queue with that stamp
        if (f.id < touched_at_[type].size()) {
            const int64_t t = __atomic_load_n(&touched_at_[type][f.id], __ATOMIC_RELAXED);
            if (t > f.at_ms && now_ms - t < reuse_quarantine_ms_) {
                freed_at_[type][f.id] = t;
                fq.push_back(Freed{f.id, t});
                continue;
            }
        }
        tab.rename(f.id, name);
        size_touched(type, f.id);
        touch(type, f.id);
        recycled_rev_[(uint64_t)type << 32 | f.id] = revision_;
        ids_recycled_++;
        if (pin) refcnt_[type][f.id] |= kPinned;
        else if (hold) refcnt_[type][f.id]++;  // (the caller's reference: released with ref(type, id, -1) once its relationships are in)
        else note_free(type, f.id);  // (unreferenced until a relationship names it: a lookup subject that never gets one is reusable again)
        return f.id;
    }
    id = tab.intern(name);
    size_touched(type, id);
    touch(type, id);
    if (!no_recycle_[type]) {
        auto &rc = refcnt_[type];
        if (rc.size() <= id) rc.resize((size_t)id + 1 + rc.size() / 2, 0);
        if (pin) rc[id] |= kPinned;
        else if (hold) rc[id]++;
        else note_free(type, id);
    }
    return id;
}

int64_t Store::now() const {
    if (now_override_) return now_override_;
    return std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

int Store::class_index(int slot, int stype, int srel, bool wildcard) const {
    auto [t, m] = schema_.slot_owner[slot];
    const auto &cls = schema_.defs[t].members[m].classes;
    for (size_t k = 0; k < cls.size(); k++)
        if (cls[k].stype == stype && cls[k].srel == srel && cls[k].wildcard == wildcard) return (int)k;
    return -1;
}

Store Store::view(int64_t at) {
    settle_all();
    Store v;
    v.schema_ = schema_;
    v.schema_loaded_ = schema_loaded_;
    v.objects_.resize(objects_.size());
    for (size_t t = 0; t < objects_.size(); t++) v.objects_[t].reserve_ids(objects_[t].count());
    // CowKeys copies share their vectors; ExpiryMap copies share their sorted bases (folded first: what is copied is a handful of delta
    // entries); the expiry index is NOT copied -- the one thing a snapshot build asks it is the validity window of its `now`, taken here
    for (auto &slot : tables_)
        for (auto &ct : slot)
            if (ct.expiry.delta_size() > 64) ct.expiry.fold();
    v.tables_ = tables_;
    v.frozen_ = true;
    v.frozen_now_ = at ? at : now();
    expiry_window(v.frozen_now_, &v.frozen_lo_, &v.frozen_hi_);
    v.wildcard_id_ = wildcard_id_;
    v.revision_ = revision_;
    v.now_override_ = now_override_;
    v.log_floor_ = revision_;
    v.bulk_revision_ = bulk_revision_;
    return v;
}

void Store::settle_all() {
    for (auto &slot : tables_)
        for (auto &ct : slot) ct.settle();
}

void ExpiryMap::fold() {
    if (delta_.empty()) return;
    std::vector<std::pair<uint64_t, int64_t>> d(delta_.begin(), delta_.end());
    std::sort(d.begin(), d.end());
    auto nb = std::make_shared<Base>();
    nb->reserve((base_ ? base_->size() : 0) + d.size());
    size_t i = 0, j = 0;
    const size_t nbase = base_ ? base_->size() : 0;
    while (i < nbase || j < d.size()) {
        if (j == d.size() || (i < nbase && (*base_)[i].first < d[j].first)) nb->push_back((*base_)[i++]);
        else {
            if (i < nbase && (*base_)[i].first == d[j].first) i++;  // (the delta's word on this key replaces the base's)
            if (d[j].second != kGone) nb->push_back(d[j]);
            j++;
        }
    }
    base_ = std::move(nb);
    delta_.clear();
}

void Store::expiry_window(int64_t now, int64_t *lo, int64_t *hi) const {
    if (frozen_) {  // a view(): the window of the `now` it was taken at; any other `now` gets the empty window [now, now + 1) -- never too wide
        if (now == frozen_now_) {
            *lo = frozen_lo_;
            *hi = frozen_hi_;
        } else {
            *lo = now;
            *hi = now + 1;
        }
        return;
    }
    *lo = LLONG_MIN;
    *hi = LLONG_MAX;
    // first entry expiring after `now`; the one before it is the last that expired at or before `now`
    auto it = expiry_index_.upper_bound(ExpiryEntry{now, INT32_MAX, INT32_MAX, UINT64_MAX});
    if (it != expiry_index_.end()) *hi = it->at;
    if (it != expiry_index_.begin()) *lo = std::prev(it)->at;
}

void Store::expiry_crossings(int64_t lo, int64_t hi, int64_t now, std::vector<Change> *out) const {
    if (now >= lo && now < hi) return;
    // the snapshot held exactly the relationships expiring at or after `hi` (nothing expires inside its window); at `now` those expiring
    // after `now` are alive: the two differ on hi <= at <= now (clock moved on) or now < at < hi (a test clock set back)
    const int64_t a = now >= hi ? hi : now + 1, b = now >= hi ? now : hi - 1;
    for (auto it = expiry_index_.lower_bound(ExpiryEntry{a, INT32_MIN, INT32_MIN, 0}); it != expiry_index_.end() && it->at <= b; ++it)
        out->push_back(Change{0, 0 /* op: the patchers look the relationship up */, it->slot, it->cls, it->key});
}

void Store::set_expiry(int slot, int cls, uint64_t key, int64_t at) {
    ClassTable &ct = tables_[slot][cls];
    int64_t old;
    if (ct.expiry.find(key, &old)) {
        if (old == at) return;
        expiry_index_.erase(ExpiryEntry{old, slot, cls, key});
        if (at) ct.expiry.set(key, at);
        else ct.expiry.erase(key);
    } else if (at) {
        ct.expiry.set(key, at);
    }
    if (at) expiry_index_.insert(ExpiryEntry{at, slot, cls, key});
}

size_t Store::gc_expired(int64_t now) {
    size_t n = 0;
    bool bumped = false;
    // (at most kGcPerWrite per call: the first write after a long idle period must not pay for a day's worth of keys -- the rest goes with the
    //  following writes; an expired relationship is dead to every reader whether or not it has been collected)
    while (n < kGcPerWrite && !expiry_index_.empty() && expiry_index_.begin()->at <= now - kGcWindowSeconds) {
        const ExpiryEntry e = *expiry_index_.begin();
        ClassTable &ct = tables_[e.slot][e.cls];
        ct.settle();
        if (ct.contains(e.key)) {
            auto &kv = ct.keys.mut();
            kv.erase(std::lower_bound(kv.begin(), kv.end(), e.key));
            ref_key(e.slot, e.cls, e.key, -1);
        }
        ct.expiry.erase(e.key);
        expiry_index_.erase(expiry_index_.begin());
        if (!bumped) {
            revision_++;
            bumped = true;
        }
        log_change(0, e.slot, e.cls, e.key);
        n++;
    }
    return n;
}

Status Store::resolve(const RelText &r, Resolved *out) {
    if (r.rtype.empty() || r.rid.empty() || r.rel.empty() || r.stype.empty() || r.sid.empty())
        return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid relationship: empty field");
    {   // API validation comes first and fails the whole request (validate.hpp): ill-formed names and ids are InvalidArgument, not "not found"
        const int vt = schema_.type_of(r.rtype), vs = schema_.type_of(r.stype);
        const bool srel_given = !r.srel.empty() && r.srel != "...";
        if ((vt < 0 && !valid_type_name(r.rtype)) || (vs < 0 && !valid_type_name(r.stype)))
            return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid relationship: object type does not match the API's pattern");
        if (((vt < 0 || schema_.defs[vt].find(r.rel) < 0) && !valid_relation_name(r.rel)) ||
            (srel_given && (vs < 0 || schema_.defs[vs].find(r.srel) < 0) && !valid_relation_name(r.srel)))
            return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid relationship: relation does not match the API's pattern");
        if (!valid_object_id(r.rid)) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid relationship: resource id `" + r.rid + "` does not match the API's pattern");
        if (r.sid != "*" && !valid_object_id(r.sid)) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid relationship: subject id `" + r.sid + "` does not match the API's pattern");
    }
    int rt = schema_.type_of(r.rtype);
    if (rt < 0) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "object definition `" + r.rtype + "` not found");
    int rl = schema_.defs[rt].find(r.rel);
    if (rl < 0) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "relation/permission `" + r.rel + "` not found under definition `" + r.rtype + "`");
    int st = schema_.type_of(r.stype);
    if (st < 0) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "object definition `" + r.stype + "` not found");
    int sr = kNoRelation;
    if (!r.srel.empty() && r.srel != "...") {
        sr = schema_.defs[st].find(r.srel);
        if (sr < 0) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "relation/permission `" + r.srel + "` not found under definition `" + r.stype + "`");
    }
    const Member &mem = schema_.defs[rt].members[rl];
    if (mem.is_permission) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "cannot write a relationship to permission `" + r.rel + "`");
    out->slot = mem.slot;
    out->rtype = rt;
    out->stype = st;
    // `T:*` is a subject class of its own (its one subject id is the name "*" in T's table); it never carries a relation
    const bool wild = r.sid == "*";
    if (wild && sr != kNoRelation) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid relationship: a wildcard subject cannot carry a relation");
    out->cls = class_index(mem.slot, st, sr, wild);
    if (out->cls < 0)
        return Status::Err(ACL_ERR_INVALID_ARGUMENT, "subjects of type `" + r.stype + (wild ? ":*" : sr == kNoRelation ? "" : "#" + r.srel) + "` are not allowed on relation `" + r.rtype + "#" + r.rel + "`");
    if (r.expires_at && !mem.classes[out->cls].expiring)
        return Status::Err(ACL_ERR_INVALID_ARGUMENT, "relation `" + r.rtype + "#" + r.rel + "` does not allow expiration for that subject type");
    // unknown objects resolve to kUnknownId: they take part in no relationship yet (Store::write gives them ids once the request is accepted)
    if (!objects_[rt].find(r.rid, &out->res)) out->res = kUnknownId;
    if (!objects_[st].find(r.sid, &out->subj)) out->subj = kUnknownId;
    out->expires = r.expires_at;
    return Status::Ok();
}

Status Store::validate_filter(const FilterText &f) const {
    if (f.rtype.empty()) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "relationship filter: resource type is required");
    {   // API validation first (validate.hpp): optional fields are checked when present
        const int vt = schema_.type_of(f.rtype), vs = f.has_stype ? schema_.type_of(f.stype) : -1;
        if ((vt < 0 && !valid_type_name(f.rtype)) || (f.has_stype && vs < 0 && !valid_type_name(f.stype)))
            return Status::Err(ACL_ERR_INVALID_ARGUMENT, "relationship filter: object type does not match the API's pattern");
        if (f.has_rel && !f.rel.empty() && (vt < 0 || schema_.defs[vt].find(f.rel) < 0) && !valid_relation_name(f.rel))
            return Status::Err(ACL_ERR_INVALID_ARGUMENT, "relationship filter: relation does not match the API's pattern");
        if (f.has_srel && !f.srel.empty() && f.srel != "..." && (vs < 0 || schema_.defs[vs].find(f.srel) < 0) && !valid_relation_name(f.srel))
            return Status::Err(ACL_ERR_INVALID_ARGUMENT, "relationship filter: subject relation does not match the API's pattern");
        if (f.has_rid && !f.rid.empty() && !valid_object_id(f.rid)) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "relationship filter: resource id does not match the API's pattern");
        if (f.has_sid && !f.sid.empty() && f.sid != "*" && !valid_object_id(f.sid))
            return Status::Err(ACL_ERR_INVALID_ARGUMENT, "relationship filter: subject id does not match the API's pattern");
    }
    int rt = schema_.type_of(f.rtype);
    if (rt < 0) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "object definition `" + f.rtype + "` not found");
    if (f.has_rel && !f.rel.empty() && schema_.defs[rt].find(f.rel) < 0)
        return Status::Err(ACL_ERR_FAILED_PRECONDITION, "relation `" + f.rel + "` not found under definition `" + f.rtype + "`");
    if (f.has_stype && schema_.type_of(f.stype) < 0) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "object definition `" + f.stype + "` not found");
    return Status::Ok();
}

void Store::scan(const FilterText &f, int64_t now, const std::function<bool(int, int, uint64_t)> &fn) {
    int rt = schema_.type_of(f.rtype);
    const Definition &d = schema_.defs[rt];
    bool want_rid = f.has_rid && !f.rid.empty();
    uint32_t rid = 0;
    if (want_rid && !objects_[rt].find(f.rid, &rid)) return;
    for (size_t m = 0; m < d.members.size(); m++) {
        const Member &mem = d.members[m];
        if (mem.is_permission) continue;
        if (f.has_rel && !f.rel.empty() && mem.name != f.rel) continue;
        for (size_t k = 0; k < mem.classes.size(); k++) {
            const SubjectClass &sc = mem.classes[k];
            bool want_sid = false;
            uint32_t sid = 0;
            if (f.has_stype) {
                if (schema_.defs[sc.stype].name != f.stype) continue;
                if (f.has_srel) {
                    bool only_none = f.srel.empty() || f.srel == "...";
                    if (only_none ? sc.srel != kNoRelation : (sc.srel == kNoRelation || schema_.defs[sc.stype].members[sc.srel].name != f.srel)) continue;
                }
                if (f.has_sid && !f.sid.empty()) {
                    want_sid = true;
                    if (!objects_[sc.stype].find(f.sid, &sid)) continue;
                }
            }
            ClassTable &ct = tables_[mem.slot][k];
            ct.settle();
            auto lo = ct.keys.begin(), hi = ct.keys.end();
            if (want_rid) {
                lo = std::lower_bound(ct.keys.begin(), ct.keys.end(), (uint64_t)rid << 32);
                hi = std::lower_bound(lo, ct.keys.end(), ((uint64_t)rid + 1) << 32);
            }
            for (auto it = lo; it != hi; ++it) {
                if (want_sid && (uint32_t)*it != sid) continue;
                if (!live(ct, *it, now)) continue;
                if (!fn(mem.slot, (int)k, *it)) return;
            }
        }
    }
}

Status Store::write(const std::vector<UpdateText> &updates, const std::vector<FilterText> &pre, uint64_t *revision) {
    if (!schema_loaded_) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    // limits pinned by the reference's engine config: pkg/spicedb/spicedb.go:35-36
    if (updates.size() > 1000) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "update count of " + std::to_string(updates.size()) + " is greater than maximum allowed of 1000");
    Status vs = validate_preconditions(pre);
    if (!vs.ok()) return vs;
    // Validation resolves WITHOUT creating ids: a rejected write (failed precondition, CREATE conflict -- the normal
    // conflict path of the dual-write workflow, workflow.go:187-201) must not grow the dense id space.
    std::vector<Resolved> rs(updates.size());
    for (size_t i = 0; i < updates.size(); i++) {
        if (updates[i].op < ACL_OP_CREATE || updates[i].op > ACL_OP_DELETE) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid update operation");
        Status s = resolve(updates[i].rel, &rs[i]);
        if (!s.ok()) return s;
        for (size_t j = 0; j < i; j++)
            if (rs[j].slot == rs[i].slot && rs[j].cls == rs[i].cls && updates[j].rel.rid == updates[i].rel.rid && updates[j].rel.sid == updates[i].rel.sid)
                return Status::Err(ACL_ERR_INVALID_ARGUMENT, "found more than one update with relationship `" + updates[i].rel.rtype + ":" + updates[i].rel.rid + "#" + updates[i].rel.rel + "` in this request");
    }
    const int64_t t = now();
    // all preconditions see the pre-write state (workflow.go:452-462)
    Status ps = eval_preconditions(pre, t);
    if (!ps.ok()) return ps;
    for (size_t i = 0; i < updates.size(); i++) {
        ClassTable &ct = tables_[rs[i].slot][rs[i].cls];
        ct.settle();
        if (updates[i].op != ACL_OP_CREATE || rs[i].res == kUnknownId || rs[i].subj == kUnknownId) continue;
        uint64_t key = (uint64_t)rs[i].res << 32 | rs[i].subj;
        if (ct.contains(key) && live(ct, key, t))
            return Status::Err(ACL_ERR_ALREADY_EXISTS, "could not CREATE relationship `" + updates[i].rel.rtype + ":" + updates[i].rel.rid + "#" + updates[i].rel.rel + "@" + updates[i].rel.stype + ":" + updates[i].rel.sid + "`, as it already existed");
    }
    // accepted: from here on nothing fails.  New objects get their ids now (a DELETE of an unknown object is a no-op).
    std::vector<char> skip(updates.size(), 0);
    for (size_t i = 0; i < updates.size(); i++) {
        if (updates[i].op == ACL_OP_DELETE) {
            skip[i] = rs[i].res == kUnknownId || rs[i].subj == kUnknownId;
            continue;
        }
    }
    // Every object this write names is HELD (one reference) until its relationships are in: a new name must not be given the id of an object
    // that is free right now but named by another update of this very request -- nor the id of a sibling interned a line earlier.
    std::vector<std::pair<int, uint32_t>> held;
    for (size_t i = 0; i < updates.size(); i++) {
        if (rs[i].res != kUnknownId) { ref(rs[i].rtype, rs[i].res, +1); held.emplace_back(rs[i].rtype, rs[i].res); }
        if (rs[i].subj != kUnknownId) { ref(rs[i].stype, rs[i].subj, +1); held.emplace_back(rs[i].stype, rs[i].subj); }
    }
    for (size_t i = 0; i < updates.size(); i++) {
        if (updates[i].op == ACL_OP_DELETE) continue;
        if (rs[i].res == kUnknownId) { rs[i].res = intern_object(rs[i].rtype, updates[i].rel.rid, false, true); held.emplace_back(rs[i].rtype, rs[i].res); }
        if (rs[i].subj == kUnknownId) { rs[i].subj = intern_object(rs[i].stype, updates[i].rel.sid, false, true); held.emplace_back(rs[i].stype, rs[i].subj); }
    }
    for (size_t i = 0; i < updates.size(); i++) {
        if (skip[i]) continue;
        ClassTable &ct = tables_[rs[i].slot][rs[i].cls];
        uint64_t key = (uint64_t)rs[i].res << 32 | rs[i].subj;
        const bool present = ct.contains(key);
        if (updates[i].op == ACL_OP_DELETE) {
            if (present) {
                auto &kv = ct.keys.mut();
                kv.erase(std::lower_bound(kv.begin(), kv.end(), key));
                ref_key(rs[i].slot, rs[i].cls, key, -1);
            }
            set_expiry(rs[i].slot, rs[i].cls, key, 0);
        } else {
            if (!present) {
                auto &kv = ct.keys.mut();
                kv.insert(std::lower_bound(kv.begin(), kv.end(), key), key);
                ref_key(rs[i].slot, rs[i].cls, key, +1);
            }
            set_expiry(rs[i].slot, rs[i].cls, key, rs[i].expires);
            {   // (a TOUCH of a relationship that is there counts too: it may have been expired, that is invisible)
                auto [ot, om] = schema_.slot_owner[rs[i].slot];
                if (tupleset_slot_[rs[i].slot] || schema_.defs[ot].members[om].classes[rs[i].cls].srel != kNoRelation) path_adds_++;
            }
        }
    }
    for (const auto &hd : held) ref(hd.first, hd.second, -1);  // (an object left without any relationship goes on its type's free list here)
    revision_++;
    for (size_t i = 0; i < updates.size(); i++)
        if (!skip[i]) log_change(updates[i].op == ACL_OP_DELETE ? ACL_OP_DELETE : ACL_OP_TOUCH, rs[i].slot, rs[i].cls, (uint64_t)rs[i].res << 32 | rs[i].subj);
    if (!expiry_index_.empty()) gc_expired(t);  // (its removals get a revision of their own: the API write's updates stay one Watch event)
    if (revision) *revision = revision_;
    return Status::Ok();
}

Status Store::validate_preconditions(const std::vector<FilterText> &pre) const {
    if (pre.size() > 1000) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "precondition count of " + std::to_string(pre.size()) + " is greater than maximum allowed of 1000");
    for (const FilterText &f : pre) {
        if (f.op != ACL_PRE_MUST_MATCH && f.op != ACL_PRE_MUST_NOT_MATCH) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid precondition operation");
        Status s = validate_filter(f);
        if (!s.ok()) return s;
    }
    return Status::Ok();
}

Status Store::eval_preconditions(const std::vector<FilterText> &pre, int64_t t) {
    for (const FilterText &f : pre) {
        bool any = false;
        scan(f, t, [&](int, int, uint64_t) {
            any = true;
            return false;
        });
        if ((f.op == ACL_PRE_MUST_MATCH && !any) || (f.op == ACL_PRE_MUST_NOT_MATCH && any))
            return Status::Err(ACL_ERR_FAILED_PRECONDITION, "unable to satisfy write precondition");
    }
    return Status::Ok();
}

Status Store::check_preconditions(const std::vector<FilterText> &pre) {
    if (!schema_loaded_) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    Status s = validate_preconditions(pre);
    return s.ok() ? eval_preconditions(pre, now()) : s;
}

Status Store::delete_by_filter(const FilterText &f, uint64_t *ndeleted, uint64_t *revision) {
    if (!schema_loaded_) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    Status s = validate_filter(f);
    if (!s.ok()) return s;
    struct Hit { int slot, cls; uint64_t key; };
    std::vector<Hit> hits;
    scan(f, now(), [&](int slot, int cls, uint64_t key) {
        hits.push_back({slot, cls, key});
        return true;
    });
    for (const Hit &h : hits) {
        ClassTable &ct = tables_[h.slot][h.cls];
        if (ct.contains(h.key)) {
            auto &kv = ct.keys.mut();
            kv.erase(std::lower_bound(kv.begin(), kv.end(), h.key));
            ref_key(h.slot, h.cls, h.key, -1);
        }
        set_expiry(h.slot, h.cls, h.key, 0);
    }
    revision_++;
    for (const Hit &h : hits) log_change(ACL_OP_DELETE, h.slot, h.cls, h.key);
    if (ndeleted) *ndeleted = hits.size();
    if (revision) *revision = revision_;
    return Status::Ok();
}

RelText Store::rel_text(int slot, int cls, uint64_t key) const {
    auto [t, m] = schema_.slot_owner[slot];
    const Member &mem = schema_.defs[t].members[m];
    const SubjectClass &sc = mem.classes[cls];
    RelText r;
    r.rtype = schema_.defs[t].name;
    const std::string *rn = objects_[t].name((uint32_t)(key >> 32));
    r.rid = rn ? *rn : "#" + std::to_string((uint32_t)(key >> 32));
    r.rel = mem.name;
    r.stype = schema_.defs[sc.stype].name;
    const std::string *sn = objects_[sc.stype].name((uint32_t)key);
    r.sid = sn ? *sn : "#" + std::to_string((uint32_t)key);
    r.srel = sc.srel == kNoRelation ? "" : schema_.defs[sc.stype].members[sc.srel].name;
    return r;
}

void Store::log_change(int op, int slot, int cls, uint64_t key) {
    if (log_.size() >= kLogCap) {
        const size_t drop = log_.size() / 2;
        log_floor_ = log_[drop - 1].revision;
        // never split one revision's updates: drop the rest of that revision too
        size_t d = drop;
        while (d < log_.size() && log_[d].revision == log_floor_) d++;
        log_.erase(log_.begin(), log_.begin() + (long)d);
        for (auto it = recycled_rev_.begin(); it != recycled_rev_.end();)  // (changes older than the floor cannot be replayed anyway)
            it = it->second <= log_floor_ ? recycled_rev_.erase(it) : std::next(it);
    }
    log_.push_back(Change{revision_, op, slot, cls, key});
}

bool Store::raw_changes_since(uint64_t after, std::vector<Change> *out) const {
    if (after < log_floor_ || after < bulk_revision_) return false;
    auto it = std::upper_bound(log_.begin(), log_.end(), after, [](uint64_t a, const Change &c) { return a < c.revision; });
    out->assign(it, log_.end());
    return true;
}

bool Store::changes_since(uint64_t after, const std::vector<int> &types, const std::function<void(const Change &, const RelText &)> &fn) const {
    if (after < log_floor_) return false;
    auto it = std::upper_bound(log_.begin(), log_.end(), after, [](uint64_t a, const Change &c) { return a < c.revision; });
    for (; it != log_.end(); ++it) {
        if (!it->op) continue;  // garbage collection of a long-expired relationship: not an API write
        const int t = schema_.slot_owner[it->slot].first;
        if (!types.empty() && std::find(types.begin(), types.end(), t) == types.end()) continue;
        if (!recycled_rev_.empty()) {  // an id that was given a new name AFTER this change: its old name is gone -- the cursor is too old to replay
            const int st = schema_.defs[t].members[schema_.slot_owner[it->slot].second].classes[it->cls].stype;
            auto a = recycled_rev_.find((uint64_t)t << 32 | (uint32_t)(it->key >> 32)), b = recycled_rev_.find((uint64_t)st << 32 | (uint32_t)it->key);
            if ((a != recycled_rev_.end() && a->second >= it->revision) || (b != recycled_rev_.end() && b->second >= it->revision)) return false;
        }
        fn(*it, rel_text(it->slot, it->cls, it->key));
    }
    return true;
}

Status Store::read(const FilterText &f, const std::function<void(const RelText &)> &cb) {
    if (!schema_loaded_) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    Status s = validate_filter(f);
    if (!s.ok()) return s;
    scan(f, now(), [&](int slot, int cls, uint64_t key) {
        RelText r = rel_text(slot, cls, key);
        int64_t at = 0;
        r.expires_at = tables_[slot][cls].expiry.find(key, &at) ? at : 0;
        cb(r);
        return true;
    });
    return Status::Ok();
}

Status Store::add_edges(int rtype, int rel, int stype, int srel, size_t n, const uint32_t *res, const uint32_t *subj) {
    if (!schema_loaded_) return Status::Err(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    if (rtype < 0 || rtype >= (int)schema_.defs.size() || stype < 0 || stype >= (int)schema_.defs.size() || rel < 0 ||
        rel >= (int)schema_.defs[rtype].members.size())
        return Status::Err(ACL_ERR_INVALID_ARGUMENT, "add_edges: type or relation index out of range");
    const Member &mem = schema_.defs[rtype].members[rel];
    if (mem.is_permission) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "add_edges: cannot write a relationship to a permission");
    no_recycle_[rtype] = no_recycle_[stype] = 1;  // caller-chosen ids: their lifetimes are the caller's business (intern_object)
    const bool wild = srel == -2;  // `stype:*` relationships: the subject is the type's wildcard id, subj[] is ignored
    int cls = class_index(mem.slot, stype, srel < 0 ? kNoRelation : srel, wild);
    if (cls < 0) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "add_edges: subject type not allowed on relation");
    ClassTable &ct = tables_[mem.slot][cls];
    for (size_t i = 0; i < n; i++)  // bit 31 of a stored subject id is the snapshot's leaf flag
        if ((res[i] | (wild ? 0u : subj[i])) & 0x80000000u) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "add_edges: object ids must be below 2^31");
    ct.pending.reserve(ct.pending.size() + n);
    uint32_t maxr = 0, maxs = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t sj = wild ? wildcard_id_[stype] : subj[i];
        ct.pending.push_back((uint64_t)res[i] << 32 | sj);
        maxr = std::max(maxr, res[i]);
        maxs = std::max(maxs, sj);
    }
    if (n) {
        objects_[rtype].reserve_ids(maxr + 1);
        objects_[stype].reserve_ids(maxs + 1);
    }
    revision_++;
    path_adds_++;
    bulk_revision_ = revision_;  // not in the change feed: snapshots older than this must be rebuilt, not patched
    return Status::Ok();
}

Status Store::load_relationship_lines(const std::string &text) {
    size_t pos = 0;
    std::vector<UpdateText> batch;
    auto flush = [&]() -> Status {
        if (batch.empty()) return Status::Ok();
        Status s = write(batch, {}, nullptr);
        batch.clear();
        return s;
    };
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(pos, e - pos);
        pos = e + 1;
        size_t b = line.find_first_not_of(" \t\r");
        if (b == std::string::npos) continue;
        size_t l = line.find_last_not_of(" \t\r");
        line = line.substr(b, l - b + 1);
        if (line.rfind("//", 0) == 0) continue;
        UpdateText u;
        u.op = ACL_OP_TOUCH;
        if (!parse_relationship_text(line, &u.rel)) return Status::Err(ACL_ERR_INVALID_ARGUMENT, "invalid relationship line `" + line + "`");
        batch.push_back(std::move(u));
        if (batch.size() == 1000) {
            Status s = flush();
            if (!s.ok()) return s;
        }
    }
    return flush();
}

}  // namespace acl
