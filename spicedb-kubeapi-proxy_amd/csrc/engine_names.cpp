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
#include "name_copies.hpp"

namespace aclint {

struct NameMirror {
    std::mutex mu;
    std::shared_mutex use;
    std::vector<NameCopyState> state;
    std::vector<uint4 *> d;  // per type; nullptr: the type has no names
    DevArray<NameTab> d_tabs;
    PinnedBuf stage;
    int device = -1;
};

void names_mirror_destroy(acl_engine *h) {
    NameMirror *m = h->name_mirror;
    if (!m) return;
    if (m->device >= 0) (void)hipSetDevice(m->device);
    for (uint4 *p : m->d)
        if (p) (void)hipFree(p);
    delete m;
    h->name_mirror = nullptr;
}

namespace {
struct MirrorOps {  // name_copies.hpp sync_name_copies on device memory
    NameMirror &m;
    hipStream_t stream;
    int resize(size_t nt) {
        for (size_t ty = nt; ty < m.d.size(); ty++)
            if (m.d[ty]) (void)hipFree(m.d[ty]);
        m.d.resize(nt, nullptr);
        return 0;
    }
    int replace(size_t ty, size_t cap, const void *bytes) {
        if (m.d[ty]) (void)hipFree(m.d[ty]);  // (hipFree waits for the device: nothing reads the old array any more)
        m.d[ty] = nullptr;
        if (!cap) return 0;
        HIP_TRY(hipMalloc((void **)&m.d[ty], cap * ObjectTable::kSlotBytes));
        HIP_TRY(hipMemcpy(m.d[ty], bytes, cap * ObjectTable::kSlotBytes, hipMemcpyHostToDevice));
        return 0;
    }
    int patch(size_t ty, const std::vector<uint32_t> &idx, const void *bytes) {
        const size_t k = idx.size(), off = (k * sizeof(uint32_t) + 63) & ~(size_t)63;
        HIP_TRY(m.stage.ensure(off + k * ObjectTable::kSlotBytes));
        std::memcpy(m.stage.p, idx.data(), k * sizeof(uint32_t));
        const char *src = static_cast<const char *>(bytes);
        for (size_t j = 0; j < k; j++) std::memcpy((char *)m.stage.p + off + j * ObjectTable::kSlotBytes, src + (size_t)idx[j] * ObjectTable::kSlotBytes, ObjectTable::kSlotBytes);
        launch_scatter_slots(stream, m.d[ty], (const uint32_t *)m.stage.dp, (const uint4 *)((const char *)m.stage.dp + off), (uint32_t)k);
        HIP_TRY(hipStreamSynchronize(stream));  // (the staging is reused by the next update; other contexts' streams must find the slots in place)
        return 0;
    }
};
}  // namespace

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
    // an array that is replaced, or copied over whole, is replaced under `use` held exclusively: no kernel of another call is reading it then
    std::unique_lock<std::shared_mutex> excl(m.use, std::defer_lock);
    if (name_copies_need_replacing(h->store, m.state) || m.d_tabs.n < nt) excl.lock();
    MirrorOps ops{m, c->stream};
    bool replaced = false;
    if (m.d.size() != m.state.size()) m.d.resize(m.state.size(), nullptr);
    int rc = sync_name_copies(h->store, m.state, ops, &replaced);
    if (rc) return rc;
    if (replaced || m.d_tabs.n < nt) {  // (only ever under the exclusive hold: name_copies_need_replacing saw it coming)
        std::vector<NameTab> tabs(nt);
        for (size_t ty = 0; ty < nt; ty++) tabs[ty] = NameTab{m.d[ty], (uint32_t)m.state[ty].cap, 0u};
        HIP_TRY(m.d_tabs.ensure(std::max<size_t>(nt, 1)));
        if (nt) HIP_TRY(hipMemcpy(m.d_tabs.p, tabs.data(), nt * sizeof(NameTab), hipMemcpyHostToDevice));
    }
    if (excl.owns_lock()) excl.unlock();
    *use_out = std::shared_lock<std::shared_mutex>(m.use);  // (taken before `mu` is given up: no replacement can slip in between)
    *tabs_out = m.d_tabs.p;
    return ACL_OK;
}

}  // namespace aclint
