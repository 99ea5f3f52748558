// engine_names.cpp -- the object-name tables mirrored in HBM, for string calls of PostFilter size (reference pkg/authz/postfilter.go:67-134: one
// CheckBulkPermissions of K list items x F checks, every item named by strings).  On the host a name costs a hash, a probe and a compare --
// 56 ns per item and thread, 0.31-0.33 ms per 65 536 items on the 16 cores a container here may use (tools/intern_bench.py), five times the
// device pass that follows.  With the slot arrays (store.hpp ObjectTable: 64-byte slots, open addressing) copied to the device, the host only
// writes each item's two names into a 64-byte record and k_resolve_names (kernels.hip) does the hashing, probing and comparing.
// VERDICT r4 next #6 asked for this; it is built, parity-green (tests/test_device_names_gpu.py) and OFF by default (ACL_DEVICE_NAMES=1):
// measured on these hosts it is no faster than the interning threads (profiles/r05_device_names.txt, engine.cpp check_bulk_strings_device).
//
// Keeping the copy current: every change of a slot moves the table's version and is remembered by the table (ObjectTable::changes); a string
// call that finds a version it has not seen brings the copy up to date before it uses it -- a handful of slots through one scatter kernel,
// or the whole array after a re-hash.  Threads: `mu` serialises the updates; `use` is held shared by every call whose kernels may still read
// the arrays and exclusively while an array is replaced or copied over whole and while the table of tables is rewritten (a single slot written
// in place needs neither: a probe that meets a half-written slot sees a tag or a name that does not match and walks on, as it would have a
// moment earlier).
#include "engine_internal.hpp"

namespace aclint {

struct NameMirror {
    std::mutex mu;
    std::shared_mutex use;
    struct PerType {
        uint4 *d = nullptr;
        size_t cap = 0;
        uint64_t version = 0;
    };
    std::vector<PerType> types;
    DevArray<NameTab> d_tabs;
    PinnedBuf stage;
    int device = -1;
};

void names_mirror_destroy(acl_engine *h) {
    NameMirror *m = h->name_mirror;
    if (!m) return;
    if (m->device >= 0) (void)hipSetDevice(m->device);
    for (auto &t : m->types)
        if (t.d) (void)hipFree(t.d);
    delete m;
    h->name_mirror = nullptr;
}

// Names lock held (shared at least).  On ACL_OK the mirror is current, *tabs_out is the device's table of tables and `use_out` holds the
// mirror shared: the caller keeps it until its stream has run dry.
int names_mirror_acquire(acl_engine *h, PassCtx *c, const NameTab **tabs_out, std::shared_lock<std::shared_mutex> *use_out) {
    {
        std::lock_guard<std::mutex> lk(h->intern_pool_mu);
        if (!h->name_mirror) h->name_mirror = new NameMirror();
    }
    NameMirror &m = *h->name_mirror;
    std::lock_guard<std::mutex> lk(m.mu);
    const size_t nt = h->store.schema().defs.size();
    m.device = c->dev->device;
    HIP_TRY(hipSetDevice(m.device));
    // what has to be replaced (under `use` held exclusively) and what can be written in place
    bool structural = m.types.size() != nt || m.d_tabs.n < nt;
    for (size_t ty = 0; ty < nt && !structural; ty++) {
        const ObjectTable &t = h->store.objects((int)ty);
        // a new array, or a re-hashed one copied over the old: a probe that ran meanwhile could miss a name that is in the table
        structural = m.types[ty].cap != t.slot_count() || (m.types[ty].version != t.version() && t.changes_are_wholesale());
    }
    std::unique_lock<std::shared_mutex> excl(m.use, std::defer_lock);
    if (structural) {
        excl.lock();  // (waits for the calls whose kernels may still be reading the arrays)
        for (size_t ty = nt; ty < m.types.size(); ty++)
            if (m.types[ty].d) (void)hipFree(m.types[ty].d);
        m.types.resize(nt);
    }
    std::vector<uint32_t> idx;
    for (size_t ty = 0; ty < nt; ty++) {
        const ObjectTable &t = h->store.objects((int)ty);
        NameMirror::PerType &p = m.types[ty];
        if (p.d && p.version == t.version()) continue;
        bool all = false;
        t.changes(&idx, &all);
        const size_t cap = t.slot_count();
        if (cap != p.cap || (!p.d && cap)) {  // (only under the exclusive hold: `structural` covers every size change)
            if (p.d) (void)hipFree(p.d);
            p.d = nullptr;
            p.cap = 0;
            if (cap) HIP_TRY(hipMalloc((void **)&p.d, cap * ObjectTable::kSlotBytes));
            p.cap = cap;
            all = true;
        }
        if (all) {
            if (cap) HIP_TRY(hipMemcpy(p.d, t.slot_bytes(), cap * ObjectTable::kSlotBytes, hipMemcpyHostToDevice));
        } else if (!idx.empty()) {
            const size_t k = idx.size(), off = (k * sizeof(uint32_t) + 63) & ~(size_t)63;
            HIP_TRY(m.stage.ensure(off + k * ObjectTable::kSlotBytes));
            std::memcpy(m.stage.p, idx.data(), k * sizeof(uint32_t));
            const char *src = static_cast<const char *>(t.slot_bytes());
            for (size_t j = 0; j < k; j++) std::memcpy((char *)m.stage.p + off + j * ObjectTable::kSlotBytes, src + (size_t)idx[j] * ObjectTable::kSlotBytes, ObjectTable::kSlotBytes);
            launch_scatter_slots(c->stream, p.d, (const uint32_t *)m.stage.dp, (const uint4 *)((const char *)m.stage.dp + off), (uint32_t)k);
            HIP_TRY(hipStreamSynchronize(c->stream));  // (the staging is reused by the next update; other contexts' streams must find the slots in place)
        }
        p.version = t.version();
    }
    if (structural) {
        std::vector<NameTab> tabs(nt);
        for (size_t ty = 0; ty < nt; ty++) tabs[ty] = NameTab{m.types[ty].d, (uint32_t)m.types[ty].cap, 0u};
        HIP_TRY(m.d_tabs.ensure(std::max<size_t>(nt, 1)));
        if (nt) HIP_TRY(hipMemcpy(m.d_tabs.p, tabs.data(), nt * sizeof(NameTab), hipMemcpyHostToDevice));
        excl.unlock();
    }
    *use_out = std::shared_lock<std::shared_mutex>(m.use);  // (taken before `mu` is given up: no replacement can slip in between)
    *tabs_out = m.d_tabs.p;
    return ACL_OK;
}

}  // namespace aclint
