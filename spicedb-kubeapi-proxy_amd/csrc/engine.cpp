// engine.cpp -- C ABI of libaclgpu.so (include/aclgpu.h): device memory, snapshot upload,
// the level loop around the frontier kernels, string <-> id plumbing.
//
// Reference behaviour mirrored at this boundary:
//   CheckBulkPermissions : pairs are index-aligned with items (pkg/authz/check.go:54-57),
//                          per-item error or permissionship (check.go:55-63)
//   LookupResources      : set of ids with HAS_PERMISSION, order irrelevant (lookups.go:85-88,129)
//   every read is fully consistent (check.go:41-46): a write is visible to the next call.
// There is no CPU evaluation path: without a GPU acl_open() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/aclgpu.h"
#include "kernels.hpp"
#include "plan.hpp"
#include "store.hpp"

using namespace acl;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
int fail(const Status &s) { return fail(s.code, s.msg); }

#define HIP_TRY(expr)                                                                                           \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return fail(ACL_ERR_INTERNAL, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
struct DevArray {
    T *p = nullptr;
    size_t n = 0;
    ~DevArray() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t ensure(size_t count) {  // grow-only, contents discarded
        if (count <= n && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const std::vector<T> &v, hipStream_t s) {
        // headroom: snapshot arrays grow when writes are patched in (plan.cpp patch_forward)
        hipError_t e = (p && v.size() <= n) ? hipSuccess : ensure(v.size() + v.size() / 4 + 16384);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
    // re-uploads elements [off, off + cnt) of v; false when v outgrew the allocation
    bool patch(const std::vector<T> &v, size_t off, size_t cnt, hipStream_t s, hipError_t *err) {
        if (!p || v.size() > n) return false;
        *err = hipMemcpyAsync(p + off, v.data() + off, cnt * sizeof(T), hipMemcpyHostToDevice, s);
        return true;
    }
};

}  // namespace

struct acl_engine {
    std::mutex mu;             // device state, snapshot, relationship tables
    std::shared_mutex names_mu;  // schema + object-name tables: shared by the callers of acl_check_one (string -> id only reads them),
                                 // exclusive (together with mu, taken after it) for everything that can add names or reload the schema
    Store store;
    Snapshot snap;
    ShardSpec shard;  // world > 1: this engine holds one shard of the graph and only the acl_shard_* entry points evaluate
    bool snap_valid = false, rev_uploaded = false;
    int device = 0;
    bool store_only = false;  // ACL_FLAG_STORE_ONLY: relationship store without a device (reads that need the GPU fail)
    hipStream_t stream = nullptr;
    int grid_blocks = 2048;
    // forward graph
    DevArray<uint32_t> d_meta, d_edges, d_buckets, d_tsb, d_tnm;
    DevArray<FwdOp> d_ops;
    DevArray<SlotProg> d_progs;
    // reverse graph
    DevArray<uint32_t> d_rmeta, d_redges, d_sbb, d_snobj, d_visited;
    DevArray<RevOp> d_rops;
    DevArray<RevProg> d_rprogs, d_rseeds;
    // frontier
    DevArray<uint4> d_fbuf[2];
    DevArray<uint32_t> d_fcounts[2], d_status;  // status = nchunks[kLevelSlots] | any[kLevelSlots] | overflow
    uint64_t frontier_entries = 0;
    uint32_t max_chunks = 0;
    uint32_t *h_status = nullptr;  // pinned
    // batch scratch
    DevArray<uint8_t> d_has, d_err, d_perm;
    DevArray<int32_t> d_errout;
    DevArray<uint4> d_items;
    uint32_t max_sub_batch = 1u << 20;
    uint32_t levels_hint = 6;
    uint32_t lk_target = 0;  // sharded lookup in flight: target slot, number of requests
    size_t lk_n = 0;
    DevArray<uint32_t> d_itemoff, d_sids;
    DevArray<uint8_t> d_keep;
    // micro-batching front-end (acl_check_one): concurrent single checks ride one device pass
    struct Waiter {
        acl_item_t item;
        uint8_t perm = 0;
        int32_t err = 0;
        int rc = 0;
        std::string msg;
        bool done = false;
        std::condition_variable cv;  // own wake-up: a finished batch does not stampede every parked caller
    };
    std::mutex q_mu;
    std::condition_variable q_cv;
    std::vector<Waiter *> queue;
    std::thread batcher;
    bool batcher_on = false, batcher_stop = false;
    uint32_t mb_max_items = 4096, mb_wait_us = 200;
    uint64_t mb_batches = 0, mb_items = 0;
    // measurement
    acl_stats_t stats{};
    bool timing = false;
    std::vector<hipEvent_t> ev;  // pairs
    size_t ev_used = 0;
    std::vector<int> ev_kind;  // per pair: 0 other, 1 expand

    DevGraph dev_graph() const {
        return DevGraph{d_meta.p, d_edges.p, d_buckets.p, d_ops.p, d_progs.p, d_tsb.p, d_tnm.p, snap.nslots, snap.ntypes, (uint32_t)snap.ops.size()};
    }
    DevFrontier dev_frontier() const {
        DevFrontier f;
        f.buf[0] = d_fbuf[0].p;
        f.buf[1] = d_fbuf[1].p;
        f.counts[0] = d_fcounts[0].p;
        f.counts[1] = d_fcounts[1].p;
        f.nchunks = d_status.p;
        f.any = d_status.p + kLevelSlots;
        f.overflow = d_status.p + 2 * kLevelSlots;  // [+1]: the level's export counter (sharded graph)
        f.nwaves = (uint32_t)grid_blocks * kWavesPerBlock;
        f.max_chunks = max_chunks;
        return f;
    }
};

namespace {

int alloc_frontier(acl_engine *h, uint64_t entries) {
    // every wave of an expand launch owns one static chunk; at least one dynamic chunk on top
    entries = std::max<uint64_t>(entries, ((uint64_t)h->grid_blocks * kWavesPerBlock + 1) * kChunk);
    uint64_t chunks = (entries + kChunk - 1) / kChunk;
    if (chunks > 0x3FFFFFu) chunks = 0x3FFFFFu;  // entry indices stay below 2^32
    for (int i = 0; i < 2; i++) {
        h->d_fbuf[i].release();
        h->d_fcounts[i].release();
        HIP_TRY(h->d_fbuf[i].ensure(chunks * kChunk));
        HIP_TRY(h->d_fcounts[i].ensure(chunks));
    }
    h->max_chunks = (uint32_t)chunks;
    h->frontier_entries = chunks * kChunk;
    return ACL_OK;
}

// ---- timing helpers: one HIP event pair per kernel launch, on the engine's stream
void ev_begin(acl_engine *h, int kind) {
    if (!h->timing) return;
    if (h->ev_used + 2 > h->ev.size()) {
        for (int i = 0; i < 2; i++) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            h->ev.push_back(e);
        }
        h->ev_kind.push_back(0);
    }
    h->ev_kind[h->ev_used / 2] = kind;
    (void)hipEventRecord(h->ev[h->ev_used], h->stream);
}
void ev_end(acl_engine *h) {
    if (!h->timing) return;
    (void)hipEventRecord(h->ev[h->ev_used + 1], h->stream);
    h->ev_used += 2;
}
void ev_collect(acl_engine *h) {  // stream must be synchronized
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) == hipSuccess) {
            h->stats.kernel_ms += ms;
            if (h->ev_kind[i / 2] == 1) h->stats.expand_ms += ms;
        }
    }
    h->ev_used = 0;
}

int ensure_snapshot(acl_engine *h) {
    if (h->store_only) return fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable");
    if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    const int64_t now = h->store.now();
    if (h->snap_valid && h->snap.revision == h->store.revision() && now >= h->snap.valid_lo && now < h->snap.valid_hi) return ACL_OK;
    // a few committed writes since the snapshot: patch the rows they touch instead of rebuilding 10 M relationships
    if (h->snap_valid && now >= h->snap.valid_lo && now < h->snap.valid_hi && h->snap.garbage_words * 4 < (h->snap.edges.size() + h->snap.buckets.size()) + 65536) {
        std::vector<Patch> patches;
        const uint64_t from_revision = h->snap.revision;
        if (patch_forward(h->store, now, &h->snap, h->shard, &patches)) {
            // the reverse rows (LookupResources), if they are on the device, follow the same feed
            bool rev_ok = h->rev_uploaded && patch_reverse(h->store, now, from_revision, &h->snap, h->shard, &patches);
            HIP_TRY(hipStreamSynchronize(h->stream));  // nothing may still be reading the rows we overwrite
            bool fits = true;
            hipError_t pe = hipSuccess;
            for (const Patch &p : patches) {
                hipError_t e1 = hipSuccess;
                switch (p.array) {
                    case Patch::META: fits = fits && h->d_meta.patch(h->snap.meta, p.off, p.n, h->stream, &e1); break;
                    case Patch::EDGES: fits = fits && h->d_edges.patch(h->snap.edges, p.off, p.n, h->stream, &e1); break;
                    case Patch::BUCKETS: fits = fits && h->d_buckets.patch(h->snap.buckets, p.off, p.n, h->stream, &e1); break;
                    case Patch::OPS: fits = fits && h->d_ops.patch(h->snap.ops, p.off, p.n, h->stream, &e1); break;
                    case Patch::RMETA: fits = fits && h->d_rmeta.patch(h->snap.rmeta, p.off, p.n, h->stream, &e1); break;
                    case Patch::REDGES: fits = fits && h->d_redges.patch(h->snap.redges, p.off, p.n, h->stream, &e1); break;
                }
                if (e1 != hipSuccess) pe = e1;
            }
            if (pe != hipSuccess) return fail(ACL_ERR_INTERNAL, std::string("snapshot patch upload: ") + hipGetErrorString(pe));
            if (!fits) {  // an array outgrew its device allocation: the host copy is already exact, upload it whole
                HIP_TRY(h->d_meta.upload(h->snap.meta, h->stream));
                HIP_TRY(h->d_edges.upload(h->snap.edges, h->stream));
                HIP_TRY(h->d_buckets.upload(h->snap.buckets, h->stream));
                HIP_TRY(h->d_ops.upload(h->snap.ops, h->stream));
                if (rev_ok) {
                    HIP_TRY(h->d_rmeta.upload(h->snap.rmeta, h->stream));
                    HIP_TRY(h->d_redges.upload(h->snap.redges, h->stream));
                }
            }
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (!rev_ok) {  // not patchable (or never built): rebuilt lazily by the next lookup
                h->rev_uploaded = false;
                h->snap.has_reverse = false;
            }
            h->stats.snapshot_patches++;
            h->stats.snapshot_edges = h->snap.nedges;
            h->stats.snapshot_edges_local = h->snap.nedges_local;
            return ACL_OK;
        }
    }
    build_forward(h->store, now, &h->snap, h->shard);
    HIP_TRY(h->d_meta.upload(h->snap.meta, h->stream));
    HIP_TRY(h->d_edges.upload(h->snap.edges, h->stream));
    HIP_TRY(h->d_buckets.upload(h->snap.buckets, h->stream));
    HIP_TRY(h->d_ops.upload(h->snap.ops, h->stream));
    HIP_TRY(h->d_progs.upload(h->snap.progs, h->stream));
    HIP_TRY(h->d_tsb.upload(h->snap.type_slot_base, h->stream));
    HIP_TRY(h->d_tnm.upload(h->snap.type_nmembers, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->snap_valid = true;
    h->rev_uploaded = false;
    h->stats.snapshot_builds++;
    h->stats.snapshot_edges = h->snap.nedges;
    h->stats.snapshot_edges_local = h->snap.nedges_local;
    h->stats.snapshot_bytes = h->snap.meta.size() * 4 + h->snap.edges.size() * 4 + h->snap.buckets.size() * 4 + h->snap.ops.size() * sizeof(FwdOp) + h->snap.progs.size() * sizeof(SlotProg);
    return ACL_OK;
}

int ensure_reverse(acl_engine *h) {
    int rc = ensure_snapshot(h);
    if (rc) return rc;
    if (h->rev_uploaded) return ACL_OK;
    build_reverse(h->store, h->store.now(), &h->snap, h->shard);
    HIP_TRY(h->d_rmeta.upload(h->snap.rmeta, h->stream));
    HIP_TRY(h->d_redges.upload(h->snap.redges, h->stream));
    HIP_TRY(h->d_rops.upload(h->snap.rops, h->stream));
    HIP_TRY(h->d_rprogs.upload(h->snap.rprogs, h->stream));
    HIP_TRY(h->d_rseeds.upload(h->snap.rseeds, h->stream));
    HIP_TRY(h->d_sbb.upload(h->snap.slot_bit_base, h->stream));
    HIP_TRY(h->d_snobj.upload(h->snap.slot_nobjects, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->rev_uploaded = true;
    h->stats.snapshot_bytes += h->snap.rmeta.size() * 4 + h->snap.redges.size() * 4;
    return ACL_OK;
}

// Runs iterations 1.. of a level loop until the frontier is empty.  `launch(iter)` enqueues
// one expansion.  Returns ACL_OK, or ACL_ERR_RESOURCE_EXHAUSTED when the frontier overflowed.
// `tail()` is enqueued after every burst, BEFORE the host learns whether the burst reached the last level: when it did
// (the common case -- the burst is sized by the previous batch's depth) the batch's epilogue has already run by the time
// the status read-back completes, instead of costing another launch + sync round trip; when it did not, the epilogue
// simply runs again after the next burst (it only reads the final has/err).
template <typename F, typename T>
int level_loop(acl_engine *h, uint32_t max_iter, F launch, uint32_t *levels_out, T tail) {
    uint32_t next = 1, burst = std::max<uint32_t>(h->levels_hint, 2);
    for (;;) {
        uint32_t last = std::min(max_iter, next + burst - 1);
        for (uint32_t it = next; it <= last; it++) {
            ev_begin(h, 1);
            launch(it);
            ev_end(h);
            h->stats.expand_launches++;
        }
        tail();
        HIP_TRY(hipMemcpyAsync(h->h_status, h->d_status.p, kStatusWords * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        ev_collect(h);
        if (h->h_status[2 * kLevelSlots] == 2) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "a relationship row exceeds the per-task enumeration limit");
        if (h->h_status[2 * kLevelSlots]) return ACL_ERR_RESOURCE_EXHAUSTED;
        uint32_t done_at = 0;
        for (uint32_t it = next; it <= last; it++)
            if (h->h_status[kLevelSlots + it] == 0) { done_at = it; break; }  // any[it]: iteration `it` produced nothing
        if (done_at || last == max_iter) {
            uint32_t lv = done_at ? done_at : max_iter;
            for (uint32_t it = 0; it < lv; it++) h->stats.frontier_entries += (uint64_t)h->h_status[it] * kChunk;  // dynamic chunks only (lower bound)
            *levels_out = lv;
            return ACL_OK;
        }
        next = last + 1;
        burst = 4;
    }
}

// one device pass over n (<= max_sub_batch) interned items already in HBM
int check_pass(acl_engine *h, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout) {
    HIP_TRY(h->d_has.ensure(h->max_sub_batch));
    HIP_TRY(h->d_err.ensure(h->max_sub_batch));
    for (int attempt = 0;; attempt++) {
        if ((uint64_t)n > h->frontier_entries) {
            int rc = alloc_frontier(h, (uint64_t)n * 4);
            if (rc) return rc;
        }
        DevGraph g = h->dev_graph();
        DevFrontier f = h->dev_frontier();
        ev_begin(h, 0);
        launch_seed(h->stream, g, f, d_items, n, h->d_has.p, h->d_err.p);  // also resets the status block
        ev_end(h);
        uint32_t levels = 0;
        int rc = level_loop(
            h, kMaxLevels, [&](uint32_t it) { launch_expand(h->stream, g, f, it, h->d_has.p, h->d_err.p); }, &levels,
            [&] {
                ev_begin(h, 0);
                launch_finalize(h->stream, n, h->d_has.p, h->d_err.p, d_perm, d_errout);
                ev_end(h);
            });
        if (rc == ACL_ERR_RESOURCE_EXHAUSTED && h->h_status[2 * kLevelSlots] == 1) {
            // frontier out of chunks: grow (up to 2^32 entries) and redo the pass
            h->stats.overflow_retries++;
            if (h->frontier_entries >= (uint64_t)0x3FFFFFu * kChunk || attempt > 8)
                return fail(ACL_ERR_RESOURCE_EXHAUSTED, "frontier capacity exceeded (" + std::to_string(h->frontier_entries) + " entries); lower max_sub_batch");
            int rc2 = alloc_frontier(h, h->frontier_entries * 4);
            if (rc2) return rc2;
            continue;
        }
        if (rc) return rc;
        h->levels_hint = levels;
        h->stats.levels_last = levels;
        h->stats.check_items += n;
        h->stats.check_passes++;
        return ACL_OK;
    }
}

int not_sharded(acl_engine *h) {
    if (h->shard.world > 1)
        return fail(ACL_ERR_FAILED_PRECONDITION, "this engine holds one shard of the graph: evaluate through acl_shard_* with the other shards");
    return ACL_OK;
}

int check_device(acl_engine *h, const uint4 *d_items, size_t n, uint8_t *d_perm, int32_t *d_errout) {
    int rc = not_sharded(h);
    if (rc) return rc;
    rc = ensure_snapshot(h);
    if (rc) return rc;
    for (size_t b = 0; b < n; b += h->max_sub_batch) {
        uint32_t m = (uint32_t)std::min<size_t>(h->max_sub_batch, n - b);
        rc = check_pass(h, d_items + b, m, d_perm + b, d_errout ? d_errout + b : nullptr);
        if (rc) return rc;
    }
    return ACL_OK;
}

bool empty(const char *s) { return !s || !*s; }

FilterText to_filter(const acl_filter_t *f) {
    FilterText o;
    o.op = f->op;
    o.rtype = f->resource_type ? f->resource_type : "";
    if (f->resource_id) { o.has_rid = true; o.rid = f->resource_id; }
    if (f->relation) { o.has_rel = true; o.rel = f->relation; }
    if (f->subject_type) { o.has_stype = true; o.stype = f->subject_type; }
    if (f->subject_id) { o.has_sid = true; o.sid = f->subject_id; }
    if (f->subject_relation) { o.has_srel = true; o.srel = f->subject_relation; }
    return o;
}

}  // namespace

extern "C" {

const char *acl_last_error(void) { return g_last_error.c_str(); }

int acl_open(const acl_config_t *cfg, acl_engine_t **out) {
    if (!out) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_open: out is NULL");
    *out = nullptr;
    if (cfg && (cfg->flags & ACL_FLAG_STORE_ONLY)) {
        auto *so = new acl_engine();
        so->store_only = true;
        so->device = -1;
        *out = so;
        return ACL_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(ACL_ERR_UNAVAILABLE, "acl_open: no HIP device available (this engine has no CPU evaluation path)");
    auto *h = new acl_engine();
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) {
        delete h;
        return fail(ACL_ERR_INVALID_ARGUMENT, "acl_open: device ordinal out of range");
    }
    h->device = dev;
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = h->d_status.ensure(kStatusWords);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_status, kStatusWords * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        std::string m = std::string("acl_open: ") + hipGetErrorString(e);
        delete h;
        return fail(ACL_ERR_UNAVAILABLE, m);
    }
    h->grid_blocks = expand_grid_blocks(dev);
    if (cfg && cfg->max_sub_batch) h->max_sub_batch = cfg->max_sub_batch;
    int rc = alloc_frontier(h, cfg && cfg->frontier_entries ? cfg->frontier_entries
                                                            : std::max<uint64_t>(16u << 20, (uint64_t)2 * h->grid_blocks * kWavesPerBlock * kChunk));
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return ACL_OK;
}

void acl_close(acl_engine_t *h) {
    if (!h) return;
    (void)acl_batcher_stop(h);
    if (h->store_only) {
        delete h;
        return;
    }
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->h_status) (void)hipHostFree(h->h_status);
    hipStream_t s = h->stream;
    delete h;
    if (s) (void)hipStreamDestroy(s);
}

int acl_load_bootstrap(acl_engine_t *h, const char *schema, size_t schema_len, const char *rels, size_t rels_len) {
    std::lock_guard<std::mutex> lk(h->mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    if (!schema) return fail(ACL_ERR_INVALID_ARGUMENT, "schema is NULL");
    Status s = h->store.load_schema(std::string(schema, schema_len));
    if (!s.ok()) return fail(s);
    h->snap_valid = false;
    if (rels && rels_len) {
        s = h->store.load_relationship_lines(std::string(rels, rels_len));
        if (!s.ok()) return fail(s);
    }
    return ACL_OK;
}

int acl_type_id(acl_engine_t *h, const char *type) {
    std::lock_guard<std::mutex> lk(h->mu);
    return type ? h->store.schema().type_of(type) : -1;
}
int acl_relation_id(acl_engine_t *h, int type, const char *name) {
    std::lock_guard<std::mutex> lk(h->mu);
    const Schema &sc = h->store.schema();
    if (!name || type < 0 || type >= (int)sc.defs.size()) return -1;
    return sc.defs[type].find(name);
}
int acl_intern(acl_engine_t *h, int type, const char *object_id, uint32_t *id_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (empty(object_id) || !id_out || type < 0 || type >= (int)sc.defs.size()) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_intern: bad argument");
    *id_out = h->store.objects(type).intern(object_id);
    return ACL_OK;
}
int acl_find(acl_engine_t *h, int type, const char *object_id, uint32_t *id_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    const Schema &sc = h->store.schema();
    if (empty(object_id) || !id_out || type < 0 || type >= (int)sc.defs.size()) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_find: bad argument");
    return h->store.objects(type).find(object_id, id_out) ? ACL_OK : fail(ACL_ERR_NOT_FOUND, "object not found");
}
const char *acl_object_name(acl_engine_t *h, int type, uint32_t id) {
    std::lock_guard<std::mutex> lk(h->mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return nullptr;
    const std::string *n = h->store.objects(type).name(id);
    return n ? n->c_str() : nullptr;
}
uint32_t acl_object_count(acl_engine_t *h, int type) {
    std::lock_guard<std::mutex> lk(h->mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return 0;
    return h->store.objects(type).count();
}

int acl_write(acl_engine_t *h, const acl_update_t *ups, int n, const acl_filter_t *pre, int m, uint64_t *rev) {
    std::lock_guard<std::mutex> lk(h->mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    if (n < 0 || m < 0 || (n && !ups) || (m && !pre)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_write: bad argument");
    std::vector<UpdateText> u(n);
    for (int i = 0; i < n; i++) {
        const acl_relationship_t &r = ups[i].rel;
        u[i].op = ups[i].op;
        u[i].rel.rtype = r.resource_type ? r.resource_type : "";
        u[i].rel.rid = r.resource_id ? r.resource_id : "";
        u[i].rel.rel = r.relation ? r.relation : "";
        u[i].rel.stype = r.subject_type ? r.subject_type : "";
        u[i].rel.sid = r.subject_id ? r.subject_id : "";
        u[i].rel.srel = r.subject_relation ? r.subject_relation : "";
        u[i].rel.expires_at = r.expires_at;
    }
    std::vector<FilterText> p(m);
    for (int i = 0; i < m; i++) p[i] = to_filter(&pre[i]);
    Status s = h->store.write(u, p, rev);
    return s.ok() ? ACL_OK : fail(s);
}

int acl_delete_by_filter(acl_engine_t *h, const acl_filter_t *f, uint64_t *n_deleted, uint64_t *rev) {
    std::lock_guard<std::mutex> lk(h->mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    if (!f) return fail(ACL_ERR_INVALID_ARGUMENT, "filter is NULL");
    Status s = h->store.delete_by_filter(to_filter(f), n_deleted, rev);
    return s.ok() ? ACL_OK : fail(s);
}

int acl_read(acl_engine_t *h, const acl_filter_t *f, acl_read_cb cb, void *user) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!f || !cb) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_read: bad argument");
    Status s = h->store.read(to_filter(f), [&](const RelText &r) {
        acl_relationship_t o{r.rtype.c_str(), r.rid.c_str(), r.rel.c_str(), r.stype.c_str(), r.sid.c_str(), r.srel.c_str(), r.expires_at};
        cb(user, &o);
    });
    return s.ok() ? ACL_OK : fail(s);
}

int acl_add_edges(acl_engine_t *h, int rtype, int rel, int stype, int srel, size_t n, const uint32_t *res, const uint32_t *subj) {
    std::lock_guard<std::mutex> lk(h->mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    Status s = h->store.add_edges(rtype, rel, stype, srel, n, res, subj);
    return s.ok() ? ACL_OK : fail(s);
}

uint64_t acl_revision(acl_engine_t *h) {
    std::lock_guard<std::mutex> lk(h->mu);
    return h->store.revision();
}
int acl_set_now(acl_engine_t *h, int64_t t) {
    std::lock_guard<std::mutex> lk(h->mu);
    h->store.set_now(t);
    return ACL_OK;
}
int acl_snapshot(acl_engine_t *h) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    return ensure_snapshot(h);
}

void *acl_stream(acl_engine_t *h) { return (void *)h->stream; }
int acl_sync(acl_engine_t *h) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ev_collect(h);
    return ACL_OK;
}

int acl_check_bulk_ids_device(acl_engine_t *h, const void *d_items, size_t n, void *d_perm_out, void *d_err_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && (!d_items || !d_perm_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_ids_device: NULL buffer");
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    return check_device(h, (const uint4 *)d_items, n, (uint8_t *)d_perm_out, (int32_t *)d_err_out);
}

int acl_check_bulk_ids(acl_engine_t *h, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && (!items || !perm_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_ids: NULL buffer");
    if (!n) return ACL_OK;
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(h->d_items.ensure(n));
    HIP_TRY(h->d_perm.ensure(n));
    HIP_TRY(h->d_errout.ensure(n));
    HIP_TRY(hipMemcpyAsync(h->d_items.p, items, n * sizeof(acl_item_t), hipMemcpyHostToDevice, h->stream));
    int rc = check_device(h, h->d_items.p, n, h->d_perm.p, h->d_errout.p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(perm_out, h->d_perm.p, n, hipMemcpyDeviceToHost, h->stream));
    if (err_out) HIP_TRY(hipMemcpyAsync(err_out, h->d_errout.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ev_collect(h);
    return ACL_OK;
}

// strings -> interned item; returns 0 or the per-item error the pair carries (check.go:55)
static int32_t intern_check_item(acl_engine_t *h, const acl_check_item_t &it, acl_item_t *out) {
    const Schema &sc = h->store.schema();
    if (empty(it.resource_type) || empty(it.resource_id) || empty(it.permission) || empty(it.subject_type) || empty(it.subject_id))
        return ACL_ERR_INVALID_ARGUMENT;  // empty request: pkg/proxy/options_test.go:101-102
    int rt = sc.type_of(it.resource_type), st = sc.type_of(it.subject_type);
    int pm = rt < 0 ? -1 : sc.defs[rt].find(it.permission);
    int sr = kNoRelation;
    bool bad = rt < 0 || st < 0 || pm < 0;
    if (!bad && !empty(it.subject_relation) && std::strcmp(it.subject_relation, "...") != 0) {
        sr = sc.defs[st].find(it.subject_relation);
        bad = sr < 0;
    }
    if (bad) return ACL_ERR_FAILED_PRECONDITION;
    // unknown object ids have no relationships: sentinels above every dense id, equal only when
    // resource and subject are the same (unknown) object
    uint32_t res, sub;
    bool kr = h->store.objects(rt).find(it.resource_id, &res), ks = h->store.objects(st).find(it.subject_id, &sub);
    if (!kr && !ks && rt == st && std::strcmp(it.resource_id, it.subject_id) == 0) res = sub = 0xFFFFFFFEu;
    else {
        if (!kr) res = 0xFFFFFFFDu;
        if (!ks) sub = 0xFFFFFFFCu;
    }
    *out = acl_item_t{(uint16_t)rt, (uint16_t)pm, res, (uint16_t)st, (uint16_t)(sr == kNoRelation ? ACL_NO_RELATION : sr), sub};
    return 0;
}

int acl_check_bulk(acl_engine_t *h, const acl_check_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out) {
    if (n && (!items || !perm_out || !err_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk: NULL buffer");
    std::vector<acl_item_t> ids;
    std::vector<size_t> where;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
        ids.reserve(n);
        where.reserve(n);
        for (size_t i = 0; i < n; i++) {
            perm_out[i] = ACL_PERM_UNSPECIFIED;
            acl_item_t o;
            err_out[i] = intern_check_item(h, items[i], &o);
            if (err_out[i]) continue;
            ids.push_back(o);
            where.push_back(i);
        }
    }
    if (ids.empty()) return ACL_OK;
    std::vector<uint8_t> p(ids.size());
    std::vector<int32_t> e(ids.size());
    int rc = acl_check_bulk_ids(h, ids.data(), ids.size(), p.data(), e.data());
    if (rc) return rc;
    for (size_t k = 0; k < ids.size(); k++) {
        perm_out[where[k]] = p[k];
        err_out[where[k]] = e[k];
    }
    return ACL_OK;
}

int acl_lookup_resources_batch(acl_engine_t *h, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n, uint32_t *bitmaps,
                               size_t words, uint64_t *counts) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && (!sids || !bitmaps)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_lookup_resources_batch: NULL buffer");
    if (h->store_only) return ensure_snapshot(h);
    int rc = not_sharded(h);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    rc = ensure_reverse(h);
    if (rc) return rc;
    const Schema &sc = h->store.schema();
    if (rtype < 0 || rtype >= (int)sc.defs.size() || stype < 0 || stype >= (int)sc.defs.size() || perm < 0 ||
        perm >= (int)sc.defs[rtype].members.size() || srel >= (int)sc.defs[stype].members.size())
        return fail(ACL_ERR_FAILED_PRECONDITION, "lookup: unknown type, permission or subject relation");
    const uint32_t target = (uint32_t)sc.slot(rtype, perm);
    const uint32_t key = sc.subject_key(stype, srel < 0 ? kNoRelation : srel);
    const uint32_t nobj = h->store.objects(rtype).count();  // (the bitmaps on the device also cover the headroom ids)
    const size_t need = (nobj + 31) / 32;
    if (words < need) return fail(ACL_ERR_INVALID_ARGUMENT, "lookup: bitmap too small (" + std::to_string(need) + " words needed)");
    const size_t vwords = (size_t)((h->snap.visited_bits + 31) / 32);
    const size_t group = std::max<size_t>(1, std::min<size_t>(n ? n : 1, ((size_t)1 << 28) / std::max<size_t>(vwords, 1)));  // <= 1 GiB of visited bits
    for (size_t b = 0; b < n; b += group) {
        const size_t m = std::min(group, n - b);
        HIP_TRY(h->d_visited.ensure(m * std::max<size_t>(vwords, 1)));
        HIP_TRY(h->d_sids.ensure(m));
        HIP_TRY(hipMemcpyAsync(h->d_sids.p, sids + b, m * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));  // pageable source: staged before return
        DevReverse r{h->d_rmeta.p, h->d_redges.p, h->d_rops.p, h->d_rprogs.p, h->d_rseeds.p, h->d_sbb.p, h->d_snobj.p, h->d_visited.p, (uint32_t)vwords};
        for (int attempt = 0;; attempt++) {
            if (m > h->frontier_entries) {
                rc = alloc_frontier(h, m * 4);
                if (rc) return rc;
            }
            DevFrontier f = h->dev_frontier();
            HIP_TRY(hipMemsetAsync(h->d_visited.p, 0, m * std::max<size_t>(vwords, 1) * 4, h->stream));
            launch_rev_seed(h->stream, f, h->d_sids.p, (uint32_t)m, key);  // seeds + status block, on the device
            uint32_t levels = 0;
            rc = level_loop(h, kMaxLevels + 1, [&](uint32_t it) { launch_rev_expand(h->stream, r, f, it); }, &levels, [&] {
                // speculative epilogue: the result rows of the target slot, one strided copy for all requests
                if (need) (void)hipMemcpy2DAsync(bitmaps + b * words, words * 4, h->d_visited.p + h->snap.slot_bit_base[target] / 32, std::max<size_t>(vwords, 1) * 4,
                                                 need * 4, m, hipMemcpyDeviceToHost, h->stream);
            });
            if (rc == ACL_ERR_RESOURCE_EXHAUSTED && h->h_status[2 * kLevelSlots] == 1) {
                h->stats.overflow_retries++;
                if (h->frontier_entries >= (uint64_t)0x3FFFFFu * kChunk || attempt > 8) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "frontier capacity exceeded in lookup");
                int rc2 = alloc_frontier(h, h->frontier_entries * 4);
                if (rc2) return rc2;
                continue;
            }
            if (rc) return rc;
            break;
        }
        for (size_t i = 0; i < m; i++) {
            uint32_t *dst = bitmaps + (b + i) * words;
            std::fill(dst + need, dst + words, 0u);
            if (counts) {
                uint64_t c = 0;
                for (size_t w = 0; w < need; w++) c += (uint64_t)__builtin_popcount(dst[w]);
                counts[b + i] = c;
            }
        }
    }
    return ACL_OK;
}

int acl_lookup_resources_ids(acl_engine_t *h, int rtype, int perm, int stype, int srel, uint32_t sid, uint32_t *bitmap, size_t words, uint64_t *count) {
    return acl_lookup_resources_batch(h, rtype, perm, stype, srel, &sid, 1, bitmap, words, count);
}

int acl_lookup_resources(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap,
                         size_t words, uint64_t *count) {
    int rt, pm, st, sr = -1;
    uint32_t sub;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        std::unique_lock<std::shared_mutex> nlk(h->names_mu);
        if (empty(rtype) || empty(perm) || empty(stype) || empty(sid)) return fail(ACL_ERR_INVALID_ARGUMENT, "invalid LookupResourcesRequest: empty field");
        if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
        const Schema &sc = h->store.schema();
        rt = sc.type_of(rtype);
        if (rt < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("object definition `") + rtype + "` not found");
        pm = sc.defs[rt].find(perm);
        if (pm < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("relation/permission `") + perm + "` not found under definition `" + rtype + "`");
        st = sc.type_of(stype);
        if (st < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("object definition `") + stype + "` not found");
        if (!empty(srel) && std::strcmp(srel, "...") != 0) {
            sr = sc.defs[st].find(srel);
            if (sr < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("relation `") + srel + "` not found under definition `" + stype + "`");
        }
        // the subject may be new to the store; give it an id so `stype:sid#srel` can be its own member
        sub = h->store.objects(st).intern(sid);
        // a subject the reverse rows have no room for (beyond the headroom ids): they are rebuilt by this lookup
        if (sr >= 0 && h->snap.has_reverse && h->store.objects(st).count() > h->snap.slot_nobjects[sc.slot(st, sr)]) {
            h->rev_uploaded = false;
            h->snap.has_reverse = false;
        }
    }
    return acl_lookup_resources_batch(h, rt, pm, st, sr, &sub, 1, bitmap, words, count);
}

int acl_stats(acl_engine_t *h, acl_stats_t *out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!out) return fail(ACL_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = h->stats;
    return ACL_OK;
}
int acl_stats_reset(acl_engine_t *h) {
    std::lock_guard<std::mutex> lk(h->mu);
    uint64_t e = h->stats.snapshot_edges, b = h->stats.snapshot_bytes, el = h->stats.snapshot_edges_local;
    uint64_t sb = h->stats.snapshot_builds, sp = h->stats.snapshot_patches;
    h->stats = acl_stats_t{};
    h->stats.snapshot_builds = sb;
    h->stats.snapshot_patches = sp;
    h->stats.snapshot_edges_local = el;
    h->stats.snapshot_edges = e;
    h->stats.snapshot_bytes = b;
    return ACL_OK;
}
int acl_set_timing(acl_engine_t *h, int on) {
    std::lock_guard<std::mutex> lk(h->mu);
    h->timing = on != 0;
    return ACL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- sharded graph
// One engine = one shard.  The host drives every shard one level at a time and moves the exported frontier
// entries between them (RCCL all-gather in aclgpu/sharded.py); nothing here talks to another GPU.
namespace {

DevShard dev_shard(acl_engine *h, void *d_export, size_t cap) {
    DevShard sh;
    sh.exp = (uint4 *)d_export;
    sh.exp_count = h->d_status.p + 2 * kLevelSlots + 1;
    sh.cap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu);
    sh.rank = h->shard.rank;
    sh.world = h->shard.world;
    return sh;
}

// reads back the status block after a level and fills the step report
int shard_report(acl_engine *h, uint32_t iter, acl_shard_step_t *out) {
    HIP_TRY(hipMemcpyAsync(h->h_status, h->d_status.p, kStatusWords * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ev_collect(h);
    out->exported = h->h_status[2 * kLevelSlots + 1];
    out->produced = h->h_status[kLevelSlots + iter];
    out->overflow = h->h_status[2 * kLevelSlots];
    return ACL_OK;
}

int shard_ready(acl_engine *h, uint32_t iter) {
    if (h->store_only) return ensure_snapshot(h);
    if (iter == 0 || iter >= kLevelSlots) return fail(ACL_ERR_INVALID_ARGUMENT, "shard step: iteration out of range");
    if (!h->snap_valid) return fail(ACL_ERR_FAILED_PRECONDITION, "shard step without acl_shard_*_begin");
    HIP_TRY(hipSetDevice(h->device));
    return ACL_OK;
}

}  // namespace

extern "C" {

int acl_shard_configure(acl_engine_t *h, uint32_t rank, uint32_t world) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (world == 0 || rank >= world) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_configure: rank must be < world");
    h->shard.rank = rank;
    h->shard.world = world;
    h->snap_valid = false;
    return ACL_OK;
}

int acl_shard_of_type(acl_engine_t *h, int type) {
    std::lock_guard<std::mutex> lk(h->mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return -1;
    return (int)shard_of_type(sc.defs[type].name, h->shard.world);
}

int acl_shard_grow_frontier(acl_engine_t *h) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    if (h->frontier_entries >= (uint64_t)0x3FFFFFu * kChunk) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "frontier capacity exceeded");
    h->stats.overflow_retries++;
    return alloc_frontier(h, h->frontier_entries * 4);
}

int acl_shard_check_begin(acl_engine_t *h, const void *d_items, size_t n, void *d_has, void *d_err) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && (!d_items || !d_has || !d_err)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_begin: NULL buffer");
    if (n > 0xFFFFFFFFu) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_begin: batch too large");
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_snapshot(h);
    if (rc) return rc;
    if ((uint64_t)n > h->frontier_entries) {
        rc = alloc_frontier(h, (uint64_t)n * 4);
        if (rc) return rc;
    }
    ev_begin(h, 0);
    launch_seed(h->stream, h->dev_graph(), h->dev_frontier(), (const uint4 *)d_items, (uint32_t)n, (uint8_t *)d_has, (uint8_t *)d_err,
                dev_shard(h, nullptr, 0));  // also resets the status block
    ev_end(h);
    h->stats.check_items += n;
    h->stats.check_passes++;
    return ACL_OK;
}

int acl_shard_check_step(acl_engine_t *h, uint32_t level, void *d_has, void *d_err, void *d_export, size_t export_cap, acl_shard_step_t *out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!out || !d_has || !d_err || (export_cap && !d_export)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_step: NULL buffer");
    int rc = shard_ready(h, level);
    if (rc) return rc;
    if (level > kMaxLevels) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_step: level beyond the dispatch depth limit");
    HIP_TRY(hipMemsetAsync(h->d_status.p + 2 * kLevelSlots + 1, 0, sizeof(uint32_t), h->stream));
    ev_begin(h, 1);
    launch_expand(h->stream, h->dev_graph(), h->dev_frontier(), level, (uint8_t *)d_has, (uint8_t *)d_err, dev_shard(h, d_export, export_cap));
    ev_end(h);
    h->stats.expand_launches++;
    h->stats.levels_last = level;
    return shard_report(h, level, out);
}

int acl_shard_check_import(acl_engine_t *h, uint32_t level, const void *d_entries, size_t n) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && !d_entries) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_import: NULL buffer");
    int rc = shard_ready(h, level);
    if (rc) return rc;
    ev_begin(h, 0);
    launch_import(h->stream, h->dev_graph(), h->dev_frontier(), level, (const uint4 *)d_entries, (uint32_t)n, dev_shard(h, nullptr, 0));
    ev_end(h);
    return ACL_OK;
}

int acl_shard_check_finish(acl_engine_t *h, const void *d_has, const void *d_err, size_t n, void *d_perm_out, void *d_err_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && (!d_has || !d_err || !d_perm_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_finish: NULL buffer");
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    ev_begin(h, 0);
    launch_finalize(h->stream, (uint32_t)n, (const uint8_t *)d_has, (const uint8_t *)d_err, (uint8_t *)d_perm_out, (int32_t *)d_err_out);
    ev_end(h);
    return ACL_OK;
}

int acl_shard_lookup_begin(acl_engine_t *h, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && !sids) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_begin: NULL buffer");
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_reverse(h);
    if (rc) return rc;
    const Schema &sc = h->store.schema();
    if (rtype < 0 || rtype >= (int)sc.defs.size() || stype < 0 || stype >= (int)sc.defs.size() || perm < 0 ||
        perm >= (int)sc.defs[rtype].members.size() || srel >= (int)sc.defs[stype].members.size())
        return fail(ACL_ERR_FAILED_PRECONDITION, "lookup: unknown type, permission or subject relation");
    const size_t vwords = std::max<size_t>((size_t)((h->snap.visited_bits + 31) / 32), 1);
    if (n * vwords > ((size_t)1 << 30)) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "lookup batch too large for one pass (visited bitmaps > 4 GiB)");
    if (n > h->frontier_entries) {
        rc = alloc_frontier(h, n * 4);
        if (rc) return rc;
    }
    h->lk_target = (uint32_t)sc.slot(rtype, perm);
    h->lk_n = n;
    const uint32_t key = sc.subject_key(stype, srel < 0 ? kNoRelation : srel);
    HIP_TRY(h->d_visited.ensure(std::max<size_t>(n, 1) * vwords));
    HIP_TRY(hipMemsetAsync(h->d_visited.p, 0, std::max<size_t>(n, 1) * vwords * 4, h->stream));
    DevFrontier f = h->dev_frontier();
    std::vector<uint4> seeds(n);
    for (size_t i = 0; i < n; i++) seeds[i] = make_uint4(sids[i], (uint32_t)i, key /* dist 0 */, 0);
    const size_t need_chunks = (n + kChunk - 1) / kChunk;
    std::vector<uint32_t> st(kStatusWords, 0), cc(std::max<size_t>(need_chunks, f.nwaves), 0);
    st[0] = need_chunks > f.nwaves ? (uint32_t)(need_chunks - f.nwaves) : 0u;
    st[kLevelSlots] = n ? 1 : 0;
    for (size_t c = 0; c < need_chunks; c++) cc[c] = (uint32_t)std::min<size_t>(kChunk, n - c * kChunk);
    HIP_TRY(hipMemcpyAsync(h->d_status.p, st.data(), st.size() * 4, hipMemcpyHostToDevice, h->stream));
    if (n) HIP_TRY(hipMemcpyAsync(f.buf[0], seeds.data(), n * sizeof(uint4), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(f.counts[0], cc.data(), cc.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ACL_OK;
}

int acl_shard_lookup_step(acl_engine_t *h, uint32_t iter, int phase, void *d_export, size_t export_cap, acl_shard_step_t *out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!out || (export_cap && !d_export)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_step: NULL buffer");
    if (phase != ACL_SHARD_VISIT && phase != ACL_SHARD_EXPAND) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_step: bad phase");
    int rc = shard_ready(h, iter);
    if (rc) return rc;
    if (!h->rev_uploaded) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_shard_lookup_step without acl_shard_lookup_begin");
    const size_t vwords = std::max<size_t>((size_t)((h->snap.visited_bits + 31) / 32), 1);
    DevReverse r{h->d_rmeta.p, h->d_redges.p, h->d_rops.p, h->d_rprogs.p, h->d_rseeds.p, h->d_sbb.p, h->d_snobj.p, h->d_visited.p, (uint32_t)vwords};
    HIP_TRY(hipMemsetAsync(h->d_status.p + 2 * kLevelSlots + 1, 0, sizeof(uint32_t), h->stream));
    ev_begin(h, 1);
    launch_rev_expand(h->stream, r, h->dev_frontier(), iter, phase == ACL_SHARD_VISIT ? REV_VISIT : REV_EXPAND, dev_shard(h, d_export, export_cap));
    ev_end(h);
    h->stats.expand_launches++;
    return shard_report(h, iter, out);
}

int acl_shard_lookup_import(acl_engine_t *h, uint32_t iter, const void *d_entries, size_t n) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (n && !d_entries) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_import: NULL buffer");
    int rc = shard_ready(h, iter);
    if (rc) return rc;
    if (!h->rev_uploaded) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_shard_lookup_import without acl_shard_lookup_begin");
    DevReverse r{h->d_rmeta.p, h->d_redges.p, h->d_rops.p, h->d_rprogs.p, h->d_rseeds.p, h->d_sbb.p, h->d_snobj.p, h->d_visited.p, 0};
    launch_rev_import(h->stream, r, h->dev_frontier(), iter, (const uint4 *)d_entries, (uint32_t)n);
    return ACL_OK;
}

int acl_shard_lookup_finish(acl_engine_t *h, void *d_bitmaps_out, size_t bitmap_words) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->store_only) return ensure_snapshot(h);
    if (!h->rev_uploaded) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_shard_lookup_finish without acl_shard_lookup_begin");
    HIP_TRY(hipSetDevice(h->device));
    const uint32_t nobj = h->store.objects(h->store.schema().slot_owner[h->lk_target].first).count();
    const size_t need = (nobj + 31) / 32;
    if (h->lk_n && (!d_bitmaps_out || bitmap_words < need))
        return fail(ACL_ERR_INVALID_ARGUMENT, "lookup: bitmap too small (" + std::to_string(need) + " words needed)");
    const size_t vwords = std::max<size_t>((size_t)((h->snap.visited_bits + 31) / 32), 1);
    const size_t woff = h->snap.slot_bit_base[h->lk_target] / 32;
    // rows of the result: only the owner of the resource type ever marks them, other shards hand back zeros
    HIP_TRY(hipMemsetAsync(d_bitmaps_out, 0, h->lk_n * bitmap_words * 4, h->stream));
    if (need)
        HIP_TRY(hipMemcpy2DAsync(d_bitmaps_out, bitmap_words * 4, h->d_visited.p + woff, vwords * 4, need * 4, h->lk_n, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ACL_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- callers either side of the kernels
extern "C" {

// filterItemsWithBulkPermissions (postfilter.go:58-182) fused: ONE bulk check of the K*F resolved pairs and the
// per-list-item AND, on the device; only K bytes come back.
static int keep_device_locked(acl_engine_t *h, const void *d_items, size_t n, const void *d_item_off, size_t k_items, void *d_keep_out) {
    HIP_TRY(h->d_perm.ensure(std::max<size_t>(n, 1)));
    int rc = check_device(h, (const uint4 *)d_items, n, h->d_perm.p, nullptr);
    if (rc) return rc;
    launch_keep(h->stream, (uint32_t)k_items, (const uint32_t *)d_item_off, h->d_perm.p, (uint8_t *)d_keep_out);
    return ACL_OK;
}

int acl_check_bulk_keep_ids_device(acl_engine_t *h, const void *d_items, size_t n, const void *d_item_off, size_t k_items, void *d_keep_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if ((n && !d_items) || (k_items && (!d_item_off || !d_keep_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_ids_device: NULL buffer");
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    return keep_device_locked(h, d_items, n, d_item_off, k_items, d_keep_out);
}

int acl_check_bulk_keep_ids(acl_engine_t *h, const acl_item_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    if ((n && !items) || (k_items && (!item_off || !keep_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_ids: NULL buffer");
    if (!k_items) return ACL_OK;
    for (size_t i = 0; i < k_items; i++)
        if (item_off[i] > item_off[i + 1] || item_off[i + 1] > n) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_ids: item_off must ascend and end within n");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->store_only) return ensure_snapshot(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(h->d_items.ensure(std::max<size_t>(n, 1)));
    HIP_TRY(h->d_itemoff.ensure(k_items + 1));
    HIP_TRY(h->d_keep.ensure(k_items));
    if (n) HIP_TRY(hipMemcpyAsync(h->d_items.p, items, n * sizeof(acl_item_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_itemoff.p, item_off, (k_items + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    int rc = keep_device_locked(h, h->d_items.p, n, h->d_itemoff.p, k_items, h->d_keep.p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(keep_out, h->d_keep.p, k_items, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ev_collect(h);
    return ACL_OK;
}

int acl_check_bulk_keep(acl_engine_t *h, const acl_check_item_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    if ((n && !items) || (k_items && (!item_off || !keep_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep: NULL buffer");
    std::vector<uint8_t> perm(std::max<size_t>(n, 1));
    std::vector<int32_t> err(std::max<size_t>(n, 1));
    int rc = acl_check_bulk(h, items, n, perm.data(), err.data());
    if (rc) return rc;
    for (size_t i = 0; i < k_items; i++) {
        if (item_off[i] > item_off[i + 1] || item_off[i + 1] > n) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep: item_off must ascend and end within n");
        bool all = true;  // pair error or anything but HAS_PERMISSION drops the item: postfilter.go:162-172
        for (uint32_t j = item_off[i]; j < item_off[i + 1]; j++) all = all && !err[j] && perm[j] == ACL_PERM_HAS_PERMISSION;
        keep_out[i] = all ? 1 : 0;
    }
    return ACL_OK;
}

// prefilterResult.IsAllowed (lookups.go:25-36) over a LookupResources bitmap instead of a set of NamespacedNames
int acl_bitmap_test_names(acl_engine_t *h, int type, const uint32_t *bitmap, size_t words, const char *const *object_ids, size_t n, uint8_t *allowed_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size() || (n && (!bitmap || !object_ids || !allowed_out)))
        return fail(ACL_ERR_INVALID_ARGUMENT, "acl_bitmap_test_names: bad argument");
    const ObjectTable &ot = h->store.objects(type);
    for (size_t i = 0; i < n; i++) {
        uint32_t id;
        allowed_out[i] = object_ids[i] && ot.find(object_ids[i], &id) && (size_t)(id >> 5) < words && ((bitmap[id >> 5] >> (id & 31u)) & 1u);
    }
    return ACL_OK;
}

// WatchService.Watch (watch.go:29-38) as a poll over the store's change feed
int acl_watch_poll(acl_engine_t *h, uint64_t after_revision, const int *types, int ntypes, acl_watch_cb cb, void *user, uint64_t *revision_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (ntypes < 0 || (ntypes && !types)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_watch_poll: bad argument");
    const Schema &sc = h->store.schema();
    std::vector<int> tv(types, types + ntypes);
    for (int t : tv)
        if (t < 0 || t >= (int)sc.defs.size()) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_watch_poll: unknown object type");
    if (revision_out) *revision_out = h->store.revision();
    if (after_revision == UINT64_MAX || !cb) return ACL_OK;  // "start from now" / cursor query only
    bool ok = h->store.changes_since(after_revision, tv, [&](const Store::Change &c, const RelText &r) {
        acl_relationship_t o{r.rtype.c_str(), r.rid.c_str(), r.rel.c_str(), r.stype.c_str(), r.sid.c_str(), r.srel.c_str(), 0};
        cb(user, c.revision, c.op, &o);
    });
    return ok ? ACL_OK : fail(ACL_ERR_OUT_OF_RANGE, "acl_watch_poll: cursor is older than the retained change feed");
}

// Test hook: brings the HOST snapshot up to date exactly as a read would (patch if possible, else rebuild) -- without
// touching a device, so it also works on a store-only engine -- and verifies it against the store.
// *patched_out = 1 when the update was a patch, 0 when it was a (re)build.
int acl_selfcheck_snapshot(acl_engine_t *h, int *patched_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    if (!h->store_only) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_selfcheck_snapshot drives the host snapshot itself: use a store-only engine");
    const int64_t now = h->store.now();
    std::vector<Patch> patches;
    bool patched = false;
    const bool current = h->snap_valid && h->snap.revision == h->store.revision() && now >= h->snap.valid_lo && now < h->snap.valid_hi;
    if (!current) {
        const uint64_t from_revision = h->snap.revision;
        if (h->snap_valid && now >= h->snap.valid_lo && now < h->snap.valid_hi) patched = patch_forward(h->store, now, &h->snap, h->shard, &patches);
        if (patched && h->snap.has_reverse && !patch_reverse(h->store, now, from_revision, &h->snap, h->shard, &patches)) h->snap.has_reverse = false;
        if (!patched) build_forward(h->store, now, &h->snap, h->shard);
        h->snap_valid = true;
    }
    if (!h->snap.has_reverse) build_reverse(h->store, now, &h->snap, h->shard);  // the hook always carries reverse rows along
    for (const Patch &p : patches) {  // every patch region must lie inside its array
        const size_t sz = p.array == Patch::META ? h->snap.meta.size() : p.array == Patch::EDGES ? h->snap.edges.size()
                        : p.array == Patch::BUCKETS ? h->snap.buckets.size() : p.array == Patch::OPS ? h->snap.ops.size()
                        : p.array == Patch::RMETA ? h->snap.rmeta.size() : h->snap.redges.size();
        if (p.off + p.n > sz) return fail(ACL_ERR_INTERNAL, "patch region outside its array");
    }
    if (patched_out) *patched_out = patched ? 1 : 0;
    std::string why;
    if (!verify_snapshot(h->store, now, h->snap, h->shard, &why)) return fail(ACL_ERR_INTERNAL, "snapshot does not match the store: " + why);
    return ACL_OK;
}

// ---- micro-batching front-end: the proxy issues many concurrent 1-item checks (check.go:76-94: one goroutine per
// check expression; watch.go:50: one per update).  acl_check_one() parks the caller, a batcher thread drains the
// queue into ONE device pass (after at most max_wait_us, or as soon as max_items are waiting) and wakes everyone.
static void batcher_loop(acl_engine_t *h) {
    std::vector<acl_engine::Waiter *> batch;
    std::vector<acl_item_t> items;
    std::vector<uint8_t> perm;
    std::vector<int32_t> err;
    bool back_to_back = false;  // the previous pass WAS the batching window: whoever arrived during it goes now
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(h->q_mu);
            if (h->queue.empty()) back_to_back = false;
            h->q_cv.wait(lk, [&] { return h->batcher_stop || !h->queue.empty(); });
            if (h->batcher_stop && h->queue.empty()) return;
            if (!back_to_back && h->queue.size() < h->mb_max_items && h->mb_wait_us)  // idle engine: let concurrent callers pile on
                h->q_cv.wait_for(lk, std::chrono::microseconds(h->mb_wait_us), [&] { return h->batcher_stop || h->queue.size() >= h->mb_max_items; });
            const size_t take = std::min<size_t>(h->queue.size(), h->mb_max_items);
            batch.assign(h->queue.begin(), h->queue.begin() + (long)take);
            h->queue.erase(h->queue.begin(), h->queue.begin() + (long)take);
        }
        items.resize(batch.size());
        perm.assign(batch.size(), 0);
        err.assign(batch.size(), 0);
        for (size_t i = 0; i < batch.size(); i++) items[i] = batch[i]->item;
        int rc = acl_check_bulk_ids(h, items.data(), items.size(), perm.data(), err.data());
        const std::string msg = rc ? acl_last_error() : "";
        {
            std::lock_guard<std::mutex> lk(h->q_mu);
            for (size_t i = 0; i < batch.size(); i++) {
                batch[i]->rc = rc;
                batch[i]->msg = msg;
                batch[i]->perm = perm[i];
                batch[i]->err = err[i];
                batch[i]->done = true;
                batch[i]->cv.notify_one();
            }
            h->mb_batches++;
            h->mb_items += batch.size();
        }
        back_to_back = true;
    }
}

int acl_batcher_start(acl_engine_t *h, uint32_t max_items, uint32_t max_wait_us) {
    std::lock_guard<std::mutex> lk(h->q_mu);
    if (h->batcher_on) return fail(ACL_ERR_FAILED_PRECONDITION, "batcher already running");
    h->mb_max_items = max_items ? max_items : 4096;
    h->mb_wait_us = max_wait_us;
    h->batcher_stop = false;
    h->batcher = std::thread(batcher_loop, h);
    h->batcher_on = true;
    return ACL_OK;
}

int acl_batcher_stop(acl_engine_t *h) {
    {
        std::lock_guard<std::mutex> lk(h->q_mu);
        if (!h->batcher_on) return ACL_OK;
        h->batcher_stop = true;
    }
    h->q_cv.notify_all();
    h->batcher.join();
    std::lock_guard<std::mutex> lk(h->q_mu);
    h->batcher_on = false;
    return ACL_OK;
}

int acl_batcher_stats(acl_engine_t *h, uint64_t *batches, uint64_t *items) {
    std::lock_guard<std::mutex> lk(h->q_mu);
    if (batches) *batches = h->mb_batches;
    if (items) *items = h->mb_items;
    return ACL_OK;
}

// CheckPermission (watch.go:50) / a 1-item CheckBulkPermissions (check.go:23-48).  Blocks until answered.
int acl_check_one(acl_engine_t *h, const acl_check_item_t *item, uint8_t *perm_out, int32_t *err_out) {
    if (!item || !perm_out || !err_out) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_one: NULL argument");
    acl_engine::Waiter w;
    {
        std::shared_lock<std::shared_mutex> nlk(h->names_mu);  // string -> id reads only: callers do not serialise on the engine
        if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
        *perm_out = ACL_PERM_UNSPECIFIED;
        *err_out = intern_check_item(h, *item, &w.item);
        if (*err_out) return ACL_OK;
    }
    {
        std::unique_lock<std::mutex> lk(h->q_mu);
        if (h->batcher_on && !h->batcher_stop) {
            h->queue.push_back(&w);
            if (h->queue.size() == 1 || h->queue.size() >= h->mb_max_items) h->q_cv.notify_one();  // only the batcher waits on q_cv
            w.cv.wait(lk, [&] { return w.done; });
            if (w.rc) return fail(w.rc, w.msg);
            *perm_out = w.perm;
            *err_out = w.err;
            return ACL_OK;
        }
    }
    return acl_check_bulk_ids(h, &w.item, 1, perm_out, err_out);  // no batcher: a device pass of its own
}

}  // extern "C"
