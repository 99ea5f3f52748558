// engine.cpp -- C ABI of libaclgpu.so (include/aclgpu.h): device memory, snapshot upload,
// the level loop around the frontier kernels, string <-> id plumbing.
//
// Reference behaviour mirrored at this boundary:
//   CheckBulkPermissions : pairs are index-aligned with items (pkg/authz/check.go:54-57),
//                          per-item error or permissionship (check.go:55-63)
//   LookupResources      : set of ids with HAS_PERMISSION, order irrelevant (lookups.go:85-88,129)
//   every read is fully consistent (check.go:41-46): a write is visible to the next call.
// There is no CPU evaluation path: without a GPU acl_open() fails.
// Threading: engine_internal.hpp (state_mu / names_mu / PassCtx pool).
#include "engine_internal.hpp"
#include "validate.hpp"

#include <pthread.h>
#include <sched.h>

namespace aclint {

thread_local std::string g_last_error;
thread_local int g_last_detail = 0;

int64_t mono_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int check_opts(const CallOpts &o) {
    if (o.cancel && *o.cancel) return fail(ACL_ERR_CANCELLED, "call cancelled by the caller");
    if (o.deadline_ns && mono_ns() >= o.deadline_ns) return fail(ACL_ERR_DEADLINE_EXCEEDED, "deadline exceeded");
    return ACL_OK;
}

PassCtx::~PassCtx() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (hipStream_t a : aux)
        if (a) {
            (void)hipStreamSynchronize(a);
            (void)hipStreamDestroy(a);
        }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    if (h_status) (void)hipHostFree(h_status);
    if (stream) (void)hipStreamDestroy(stream);
}

int alloc_frontier(acl_engine *h, PassCtx *c, uint64_t entries) {
    // every wave of an expand launch owns one static chunk; at least one dynamic chunk on top
    entries = std::max<uint64_t>(entries, ((uint64_t)c->dev->grid_blocks * kWavesPerBlock + 1) * kChunk);
    uint64_t chunks = (entries + kChunk - 1) / kChunk;
    if (chunks > kMaxFrontierChunks) chunks = kMaxFrontierChunks;  // byte offsets of entries stay below 2^32 (kernels.hip gld / gst)
    for (int i = 0; i < 2; i++) {
        c->d_fbuf[i].release();
        c->d_fcounts[i].release();
        HIP_TRY(c->d_fbuf[i].ensure(chunks * kChunk));
        HIP_TRY(c->d_fcounts[i].ensure(chunks));
    }
    c->max_chunks = (uint32_t)chunks;
    c->frontier_entries = chunks * kChunk;
    return ACL_OK;
}

int new_ctx(acl_engine *h, DevState *d, std::unique_ptr<PassCtx> *out, int index) {
    HIP_TRY(hipSetDevice(d->device));
    auto c = std::make_unique<PassCtx>();
    c->index = index;
    c->dev = d;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(c->d_status.ensure(kStatusWords));
    HIP_TRY(c->d_done.ensure(1));
    HIP_TRY(hipMemset(c->d_done.p, 0, sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&c->h_status, kStatusWords * sizeof(uint32_t), hipHostMallocDefault));
    int rc = alloc_frontier(h, c.get(),
                            h->cfg_frontier_entries ? h->cfg_frontier_entries : std::max<uint64_t>(32u << 20, (uint64_t)2 * d->grid_blocks * kWavesPerBlock * kChunk));  // 2 x 512 MiB: the single-launch walk carves its blocks' private regions out of these
    if (rc) return rc;
    *out = std::move(c);
    return ACL_OK;
}

// ---- timing helpers: one HIP event pair per kernel launch, on the context's stream
void ev_begin(PassCtx *c, int kind) {
    if (!c->timing) return;
    if (c->ev_used + 2 > c->ev.size()) {
        for (int i = 0; i < 2; i++) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            c->ev.push_back(e);
        }
        c->ev_kind.push_back(0);
    }
    c->ev_kind[c->ev_used / 2] = kind;
    (void)hipEventRecord(c->ev[c->ev_used], c->stream);
}
void ev_end(PassCtx *c) {
    if (!c->timing) return;
    (void)hipEventRecord(c->ev[c->ev_used + 1], c->stream);
    c->ev_used += 2;
}
void ev_collect(PassCtx *c) {  // stream must be synchronized
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) {
            c->stats.kernel_ms += ms;
            if (c->ev_kind[i / 2] == 1) c->stats.expand_ms += ms;
            else if (c->ev_kind[i / 2] == 2) c->stats.local_ms += ms;
            else if (c->ev_kind[i / 2] == 3) c->stats.rev_local_ms += ms;
        }
    }
    c->ev_used = 0;
}

void merge_stats(acl_engine *h, PassCtx *c) {
    std::lock_guard<std::mutex> lk(h->stats_mu);
    acl_stats_t &a = h->stats, &b = c->stats;
    a.check_items += b.check_items;
    a.check_passes += b.check_passes;
    a.expand_launches += b.expand_launches;
    if (b.check_passes || b.lookup_requests) a.levels_last = b.levels_last;
    a.frontier_entries += b.frontier_entries;
    a.kernel_ms += b.kernel_ms;
    a.expand_ms += b.expand_ms;
    a.local_ms += b.local_ms;
    a.local_passes += b.local_passes;
    a.rev_local_ms += b.rev_local_ms;
    a.rev_local_passes += b.rev_local_passes;
    a.lookup_requests += b.lookup_requests;
    a.overflow_retries += b.overflow_retries;
    b = acl_stats_t{};
}

bool snapshot_current(acl_engine *h, bool need_reverse) {
    if (!h->snap_valid || !h->all_dev_valid() || h->snap.revision != h->store.revision()) return false;
    const int64_t now = h->store.now();
    if (now < h->snap.valid_lo || now >= h->snap.valid_hi) return false;
    return !need_reverse || h->all_rev_uploaded();
}

// Uploads the regions a patch touched.  A region is one hipMemcpyAsync (a few microseconds of API time each): past a few
// hundred regions of one array the whole array goes instead (70 MB of snapshot cross PCIe in ~2 ms; 6 000 regions took 28 ms).
// *fits = false when a host array outgrew its device allocation (the caller uploads everything).
struct SnapArrays {
    DevArray<uint32_t> *meta, *edges, *buckets, *rmeta, *redges;
    DevArray<FwdOp> *ops;
};
static hipError_t upload_patches(const Snapshot &sn, const std::vector<Patch> &patches, const SnapArrays &a, bool with_reverse, hipStream_t s, bool *fits) {
    size_t cnt[6] = {0, 0, 0, 0, 0, 0};
    for (const Patch &p : patches) cnt[p.array]++;
    constexpr size_t kWhole = 256;
    hipError_t err = hipSuccess;
    auto whole = [&](auto *dev, const auto &host) {
        if (!dev->p || host.size() > dev->n) *fits = false;
        else if (hipError_t e = hipMemcpyAsync(dev->p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice, s); e != hipSuccess) err = e;
    };
    if (cnt[Patch::META] > kWhole) whole(a.meta, sn.meta);
    if (cnt[Patch::EDGES] > kWhole) whole(a.edges, sn.edges);
    if (cnt[Patch::BUCKETS] > kWhole) whole(a.buckets, sn.buckets);
    if (with_reverse && cnt[Patch::RMETA] > kWhole) whole(a.rmeta, sn.rmeta);
    if (with_reverse && cnt[Patch::REDGES] > kWhole) whole(a.redges, sn.redges);
    for (const Patch &p : patches) {
        if (cnt[p.array] > kWhole && p.array != Patch::OPS) continue;
        hipError_t e1 = hipSuccess;
        switch (p.array) {
            case Patch::META: *fits = *fits && a.meta->patch(sn.meta, p.off, p.n, s, &e1); break;
            case Patch::EDGES: *fits = *fits && a.edges->patch(sn.edges, p.off, p.n, s, &e1); break;
            case Patch::BUCKETS: *fits = *fits && a.buckets->patch(sn.buckets, p.off, p.n, s, &e1); break;
            case Patch::OPS: *fits = *fits && a.ops->patch(sn.ops, p.off, p.n, s, &e1); break;
            case Patch::RMETA: if (with_reverse) *fits = *fits && a.rmeta->patch(sn.rmeta, p.off, p.n, s, &e1); break;
            case Patch::REDGES: if (with_reverse) *fits = *fits && a.redges->patch(sn.redges, p.off, p.n, s, &e1); break;
        }
        if (e1 != hipSuccess) err = e1;
    }
    return err;
}

// ---- background compaction (engine_internal.hpp Compaction); everything here runs under state_mu EXCLUSIVE except the worker
static bool compaction_due(acl_engine *h) {
    const Snapshot &s = h->snap;
    if (s.garbage_words * 8 > s.edges.size() + s.buckets.size() + h->compaction_slack) return true;  // half of the 25 % that forces a rebuild
    const Schema &sc = h->store.schema();
    // a table's spare ids running low: fewer left than a tenth of the table, or than 8 192 (half of the smallest headroom) -- the build
    // must finish before they are gone, and a small table of a fast-growing type (lock / workflow / activity ids) has no "last 10 %" to speak of
    auto low = [](uint64_t used, uint64_t cap) { return cap && used + std::max<uint64_t>(cap / 10, 8192) > cap; };
    for (int slot = 0; slot < sc.nslots && slot < (int)s.lay.size(); slot++) {
        const RelLayout &l = s.lay[slot];
        if (low(h->store.objects(sc.slot_owner[slot].first).count(), l.nrows)) return true;
        for (size_t k = 0; k < l.cls.size(); k++) {
            const auto [t, m] = sc.slot_owner[slot];
            if (l.cls[k].hashed && !sc.defs[t].members[m].classes[k].wildcard && low(h->store.objects(sc.defs[t].members[m].classes[k].stype).count(), l.cls[k].nsubjects)) return true;
        }
    }
    return false;
}

static void compaction_start(acl_engine *h) {
    if (!h->compaction_enabled || h->store_only) return;
    if (!h->compaction) h->compaction = std::make_unique<Compaction>();
    Compaction *c = h->compaction.get();
    // one at a time -- and a FINISHED build (2) waits for the next reader to adopt it: starting another one here threw it away.  That was not
    // rare: the worker's last uploads and this thread's patch upload meet in the runtime, so the build tended to finish exactly between this
    // read's adoption check and this call (4 of 5 builds were dropped in the dual-write run, and the tables ran out of spare ids meanwhile).
    if (c->state.load() == 1 || c->state.load() == 2) return;
    if (getenv("ACL_DEBUG_REBUILD")) fprintf(stderr, "[aclgpu] background build starts at revision %llu (previous state %d)\n", (unsigned long long)h->store.revision(), c->state.load());
    if (c->worker.joinable()) c->worker.join();
    // one set of fresh arrays + an upload stream per replica (created on that replica's device)
    while (c->per.size() < h->devs.size()) c->per.push_back(std::make_unique<Compaction::PerDevice>());
    for (size_t i = 0; i < h->devs.size(); i++) {
        Compaction::PerDevice &pd = *c->per[i];
        pd.device = h->devs[i]->device;
        if (!pd.stream && (hipSetDevice(pd.device) != hipSuccess || hipStreamCreateWithFlags(&pd.stream, hipStreamNonBlocking) != hipSuccess)) return;
    }
    c->now = h->store.now();
    auto view = std::make_shared<Store>(h->store.view(c->now));  // tables shared copy-on-write, expiry maps share their sorted bases: O(#tables), not O(#relationships) or O(#expiring keys)
    c->shard = h->shard;
    c->with_reverse = h->all_rev_uploaded();
    c->error.clear();
    c->state.store(1);
    const size_t ndev = h->devs.size();
    c->worker = std::thread([c, view, ndev] {
        // ONE host build, uploaded to every replica
        build_forward(*view, c->now, &c->snap, c->shard);
        if (c->with_reverse) build_reverse(*view, c->now, &c->snap, c->shard);
        const Snapshot &s = c->snap;
        bool ok = std::max({s.meta.size(), s.edges.size(), s.buckets.size()}) < ((size_t)1 << 30);
        for (size_t i = 0; ok && i < ndev; i++) {
            Compaction::PerDevice &pd = *c->per[i];
            auto up = [&](auto &dev, const auto &host) { return dev.upload(host, pd.stream) == hipSuccess; };
            ok = hipSetDevice(pd.device) == hipSuccess && up(pd.d_meta, s.meta) && up(pd.d_edges, s.edges) && up(pd.d_buckets, s.buckets) && up(pd.d_ops, s.ops) &&
                 up(pd.d_progs, s.progs) && up(pd.d_bexpr, s.bexpr) && up(pd.d_tsb, s.type_slot_base) && up(pd.d_tnm, s.type_nmembers);
            if (ok && c->with_reverse)
                ok = up(pd.d_rmeta, s.rmeta) && up(pd.d_redges, s.redges) && up(pd.d_rops, s.rops) && up(pd.d_rprogs, s.rprogs) && up(pd.d_rseeds, s.rseeds) && up(pd.d_rdest, s.rdest) &&
                     up(pd.d_sbb, s.slot_bit_base) && up(pd.d_snobj, s.slot_nobjects);
        }
        for (size_t i = 0; ok && i < ndev; i++) ok = hipSetDevice(c->per[i]->device) == hipSuccess && hipStreamSynchronize(c->per[i]->stream) == hipSuccess;
        c->state.store(ok ? 2 : 3);
    });
}

void compaction_join(acl_engine *h) {
    if (!h->compaction) return;
    if (h->compaction->worker.joinable()) h->compaction->worker.join();
    h->compaction->state.store(0);
}

static void refresh_local_blocks(acl_engine *h) {  // (the single-launch kernel's LDS depends on the schema)
    for (auto &d : h->devs) {
        d->local_blocks = local_grid_blocks(d->device, (h->snap.progs.size() + h->snap.ops.size()) * 32);
        d->local_blocks_wide = local_grid_blocks(d->device, (h->snap.progs.size() + h->snap.ops.size()) * 32, true);
        if (const char *ev = getenv("ACL_LOCAL_BLOCKS_WIDE")) d->local_blocks_wide = std::max(1, atoi(ev));  // A/B knob: resident blocks the wide walk plans its units for
    }
}

// A finished build: bring it from the view's revision to the store's with the ordinary patcher, then swap it in on every replica.
// Returns true when the engine's snapshot is now the compacted one (and current).
static bool compaction_adopt(acl_engine *h, int64_t now) {
    Compaction *c = h->compaction.get();
    if (!c || c->state.load() != 2) {
        if (c && c->state.load() == 3) {
            if (getenv("ACL_DEBUG_REBUILD")) fprintf(stderr, "[aclgpu] background build failed: %s\n", c->error.c_str());
            c->state.store(0);
        }
        return false;
    }
    c->state.store(0);
    if (c->worker.joinable()) c->worker.join();
    if (c->shard.rank != h->shard.rank || c->shard.world != h->shard.world || c->per.size() < h->devs.size()) return false;
    std::vector<Patch> patches;
    const uint64_t from = c->snap.revision;
    if (getenv("ACL_DEBUG_REBUILD")) fprintf(stderr, "[aclgpu] adopting the background build of revision %llu at revision %llu\n", (unsigned long long)from, (unsigned long long)h->store.revision());
    if (!patch_forward(h->store, now, &c->snap, h->shard, &patches, (size_t)1 << 19)) return false;  // (a bulk load meanwhile: the synchronous path decides)
    bool rev_ok = c->with_reverse && patch_reverse(h->store, now, from, &c->snap, h->shard, &patches);
    for (size_t i = 0; i < h->devs.size(); i++) {  // the catch-up patch reaches every replica's fresh arrays before any of them is swapped in
        Compaction::PerDevice &pd = *c->per[i];
        bool fits = true;
        if (hipSetDevice(pd.device) != hipSuccess) return false;
        const hipError_t pe = upload_patches(c->snap, patches, SnapArrays{&pd.d_meta, &pd.d_edges, &pd.d_buckets, &pd.d_rmeta, &pd.d_redges, &pd.d_ops}, rev_ok, h->devs[i]->up_stream, &fits);
        if (pe != hipSuccess || !fits || hipStreamSynchronize(h->devs[i]->up_stream) != hipSuccess) return false;
    }
    // swap: the old arrays go to the compaction object and are freed (or reused) by its next run
    h->set_dev_valid(false);
    h->snap = std::move(c->snap);
    c->snap = Snapshot();
    for (size_t i = 0; i < h->devs.size(); i++) {
        DevState &d = *h->devs[i];
        Compaction::PerDevice &pd = *c->per[i];
        d.d_meta.swap(pd.d_meta);
        d.d_edges.swap(pd.d_edges);
        d.d_buckets.swap(pd.d_buckets);
        d.d_ops.swap(pd.d_ops);
        d.d_progs.swap(pd.d_progs);
        d.d_bexpr.swap(pd.d_bexpr);
        d.d_tsb.swap(pd.d_tsb);
        d.d_tnm.swap(pd.d_tnm);
        if (rev_ok) {
            d.d_rmeta.swap(pd.d_rmeta);
            d.d_redges.swap(pd.d_redges);
            d.d_rops.swap(pd.d_rops);
            d.d_rprogs.swap(pd.d_rprogs);
            d.d_rseeds.swap(pd.d_rseeds);
            d.d_rdest.swap(pd.d_rdest);
            d.d_sbb.swap(pd.d_sbb);
            d.d_snobj.swap(pd.d_snobj);
        }
        d.rev_uploaded = rev_ok;
        d.dev_valid = true;
    }
    if (!rev_ok) h->snap.has_reverse = false;
    h->snap_valid = true;
    refresh_local_blocks(h);
    std::lock_guard<std::mutex> lk(h->stats_mu);
    h->stats.snapshot_compactions++;
    h->stats.snapshot_edges = h->snap.nedges;
    h->stats.snapshot_edges_local = h->snap.nedges_local;
    h->stats.snapshot_bytes = h->snap.meta.size() * 4 + h->snap.edges.size() * 4 + h->snap.buckets.size() * 4 + h->snap.ops.size() * sizeof(FwdOp) +
                              h->snap.progs.size() * sizeof(SlotProg) + (rev_ok ? h->snap.rmeta.size() * 4 + h->snap.redges.size() * 4 : 0);
    return true;
}

// caller holds state_mu EXCLUSIVE: no evaluation is reading the device arrays of ANY replica.  Whatever changes the snapshot here reaches
// every replica before the function returns (DevState): an evaluation that starts afterwards answers for the store as it is now, whichever
// device it lands on -- the reference's one client is read-your-writes for the whole process (check.go:41-46, activity.go:60-77).
int ensure_snapshot(acl_engine *h) {
    if (h->store_only) return fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable");
    if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    if (snapshot_current(h, false)) return ACL_OK;
    h->snap_epoch++;  // (whatever happens below changes the snapshot or fails: what was learnt about the old one -- acl_engine::deep_known -- is not carried over)
    // (schemas with `&` / `-` on a sharded graph: the snapshot builds like any other; which entry points evaluate it is ShardCall::begin's business)
    const int64_t now = h->store.now();
    if (h->snap_valid && h->all_dev_valid() && compaction_adopt(h, now) && snapshot_current(h, false)) return ACL_OK;  // a background rebuild finished: swap it in
    // a few committed writes since the snapshot: patch the rows they touch instead of rebuilding 10 M relationships
    // (... or an expiration passed: the relationships that ran out are part of the patcher's feed)
    if (h->snap_valid && h->all_dev_valid() &&
        h->snap.garbage_words * 4 < (h->snap.edges.size() + h->snap.buckets.size()) + 65536) {
        std::vector<Patch> patches;
        const uint64_t from_revision = h->snap.revision;
        if (patch_forward(h->store, now, &h->snap, h->shard, &patches)) {
            // the reverse rows (LookupResources), if they are on the devices, follow the same feed
            const bool had_rev = h->all_rev_uploaded();
            bool rev_ok = had_rev && patch_reverse(h->store, now, from_revision, &h->snap, h->shard, &patches);
            h->set_dev_valid(false);  // until every region below has reached every device
            h->set_rev_uploaded(false);
            if (std::max({h->snap.meta.size(), h->snap.edges.size(), h->snap.buckets.size()}) >= ((size_t)1 << 30))
                return fail(ACL_ERR_RESOURCE_EXHAUSTED, "snapshot array beyond 4 GiB (more than ~1 G relationships in one array): shard the graph (acl_shard_configure)");
            for (auto &dp : h->devs) {
                DevState &d = *dp;
                hipStream_t s = d.up_stream;
                HIP_TRY(hipSetDevice(d.device));
                bool fits = true;
                const hipError_t pe = upload_patches(h->snap, patches, SnapArrays{&d.d_meta, &d.d_edges, &d.d_buckets, &d.d_rmeta, &d.d_redges, &d.d_ops}, rev_ok, s, &fits);
                if (pe != hipSuccess) return fail(ACL_ERR_INTERNAL, std::string("snapshot patch upload: ") + hipGetErrorString(pe));
                if (!fits) {  // an array outgrew its device allocation: the host copy is already exact, upload it whole
                    HIP_TRY(d.d_meta.upload(h->snap.meta, s));
                    HIP_TRY(d.d_edges.upload(h->snap.edges, s));
                    HIP_TRY(d.d_buckets.upload(h->snap.buckets, s));
                    HIP_TRY(d.d_ops.upload(h->snap.ops, s));
                    if (rev_ok) {
                        HIP_TRY(d.d_rmeta.upload(h->snap.rmeta, s));
                        HIP_TRY(d.d_redges.upload(h->snap.redges, s));
                    }
                }
            }
            for (auto &dp : h->devs) {  // (the replicas' uploads overlap; one wait each)
                HIP_TRY(hipSetDevice(dp->device));
                HIP_TRY(hipStreamSynchronize(dp->up_stream));
                dp->dev_valid = true;
                dp->rev_uploaded = had_rev && rev_ok;
            }
            if (!(had_rev && rev_ok)) h->snap.has_reverse = false;  // not patchable (or never built): rebuilt lazily by the next lookup
            {
                std::lock_guard<std::mutex> lk(h->stats_mu);
                h->stats.snapshot_patches++;
                h->stats.snapshot_edges = h->snap.nedges;
                h->stats.snapshot_edges_local = h->snap.nedges_local;
            }
            if (compaction_due(h)) compaction_start(h);  // garbage / headroom half used: build the next snapshot in the background
            return ACL_OK;
        }
    }
    if (getenv("ACL_DEBUG_REBUILD"))
        fprintf(stderr, "[aclgpu] synchronous rebuild: snap_valid=%d dev_valid=%d window=[%lld,%lld) now=%lld garbage=%llu of %zu store_rev=%llu snap_rev=%llu\n",
                (int)h->snap_valid, (int)h->all_dev_valid(), (long long)h->snap.valid_lo, (long long)h->snap.valid_hi, (long long)now,
                (unsigned long long)h->snap.garbage_words, h->snap.edges.size() + h->snap.buckets.size(), (unsigned long long)h->store.revision(),
                (unsigned long long)h->snap.revision);
    h->snap_valid = false;
    h->set_dev_valid(false);
    h->set_rev_uploaded(false);
    build_forward(h->store, now, &h->snap, h->shard);
    h->snap_valid = true;
    // the kernels address every snapshot array as base + 32-bit byte offset (kernels.hip gld): refuse what does not fit
    if (std::max({h->snap.meta.size(), h->snap.edges.size(), h->snap.buckets.size()}) >= ((size_t)1 << 30))
        return fail(ACL_ERR_RESOURCE_EXHAUSTED, "snapshot array beyond 4 GiB (more than ~1 G relationships in one array): shard the graph (acl_shard_configure)");
    for (auto &dp : h->devs) {
        DevState &d = *dp;
        hipStream_t s = d.up_stream;
        HIP_TRY(hipSetDevice(d.device));
        HIP_TRY(d.d_meta.upload(h->snap.meta, s));
        HIP_TRY(d.d_edges.upload(h->snap.edges, s));
        HIP_TRY(d.d_buckets.upload(h->snap.buckets, s));
        HIP_TRY(d.d_ops.upload(h->snap.ops, s));
        HIP_TRY(d.d_progs.upload(h->snap.progs, s));
        HIP_TRY(d.d_bexpr.upload(h->snap.bexpr, s));
        HIP_TRY(d.d_tsb.upload(h->snap.type_slot_base, s));
        HIP_TRY(d.d_tnm.upload(h->snap.type_nmembers, s));
    }
    for (auto &dp : h->devs) {
        HIP_TRY(hipSetDevice(dp->device));
        HIP_TRY(hipStreamSynchronize(dp->up_stream));
        dp->dev_valid = true;
    }
    refresh_local_blocks(h);
    std::lock_guard<std::mutex> lk(h->stats_mu);
    h->stats.snapshot_builds++;
    h->walk_no_direct.store(false, std::memory_order_relaxed);  // (a new snapshot: the direct task lists get another chance)
    h->stats.snapshot_edges = h->snap.nedges;
    h->stats.snapshot_edges_local = h->snap.nedges_local;
    h->stats.snapshot_bytes = h->snap.meta.size() * 4 + h->snap.edges.size() * 4 + h->snap.buckets.size() * 4 + h->snap.ops.size() * sizeof(FwdOp) + h->snap.progs.size() * sizeof(SlotProg);
    return ACL_OK;
}

// do the reverse rows' visited bitmaps cover every id a walk could mark?  Ids are interned without a revision bump
// (acl_intern, a lookup's subject, a LookupResources on a new object), so "same revision" does not imply it.
static bool reverse_covers(acl_engine *h) {
    const Schema &sc = h->store.schema();
    if (h->snap.slot_nobjects.size() != (size_t)sc.nslots) return false;
    for (int slot = 0; slot < sc.nslots; slot++)
        if (h->store.objects(sc.slot_owner[slot].first).count() > h->snap.slot_nobjects[slot]) return false;
    return true;
}

int ensure_reverse(acl_engine *h) {
    int rc = ensure_snapshot(h);
    if (rc) return rc;
    if (h->all_rev_uploaded() && reverse_covers(h)) return ACL_OK;
    h->set_rev_uploaded(false);
    build_reverse(h->store, h->store.now(), &h->snap, h->shard);
    for (auto &dp : h->devs) {
        DevState &d = *dp;
        hipStream_t s = d.up_stream;
        HIP_TRY(hipSetDevice(d.device));
        HIP_TRY(d.d_rmeta.upload(h->snap.rmeta, s));
        HIP_TRY(d.d_redges.upload(h->snap.redges, s));
        HIP_TRY(d.d_rops.upload(h->snap.rops, s));
        HIP_TRY(d.d_rprogs.upload(h->snap.rprogs, s));
        HIP_TRY(d.d_rseeds.upload(h->snap.rseeds, s));
        HIP_TRY(d.d_rdest.upload(h->snap.rdest, s));
        HIP_TRY(d.d_sbb.upload(h->snap.slot_bit_base, s));
        HIP_TRY(d.d_snobj.upload(h->snap.slot_nobjects, s));
    }
    for (auto &dp : h->devs) {
        HIP_TRY(hipSetDevice(dp->device));
        HIP_TRY(hipStreamSynchronize(dp->up_stream));
        dp->rev_uploaded = true;
    }
    std::lock_guard<std::mutex> lk(h->stats_mu);
    h->stats.snapshot_bytes += h->snap.rmeta.size() * 4 + h->snap.redges.size() * 4;
    return ACL_OK;
}

// device ordinal a device pointer lives on (-1: one replica, or not a device pointer: any replica will do)
int device_of(acl_engine *h, const void *p) {
    if (h->devs.size() < 2 || !p) return -1;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return a.device;
}

int Eval::begin(acl_engine *h_, bool need_reverse, const CallOpts &opts, int rev_key_slot, int on_device) {
    h = h_;
    if (h->store_only) return fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable");
    int rc = check_opts(opts);
    if (rc) return rc;
    for (;;) {
        h->state_mu.lock_shared();
        bool ok = snapshot_current(h, need_reverse);
        // a `type#relation` lookup subject is itself a state of the walk: its id must lie inside the visited bitmap
        if (ok && need_reverse && rev_key_slot >= 0 &&
            h->store.objects(h->store.schema().slot_owner[rev_key_slot].first).count() > h->snap.slot_nobjects[rev_key_slot])
            ok = false;
        if (ok) {
            locked = true;
            break;
        }
        h->state_mu.unlock_shared();
        std::lock_guard<RwLock> lk(h->state_mu);
        rc = need_reverse ? ensure_reverse(h) : ensure_snapshot(h);
        if (rc) return rc;
    }
    // a context from the pool (created on demand up to max_ctx per replica)
    std::unique_lock<std::mutex> lk(h->pool_mu);
    for (;;) {
        // Replicas (engines opened on several devices): the least loaded one that can serve the call -- ties go round the devices, so N
        // blocking callers end up on N devices.  Within a replica: the context released last (its buffers are the warmest).
        DevState *bd = nullptr;
        int bpick = -1;
        bool bcreate = false;
        const size_t nd = h->devs.size(), d0 = nd > 1 ? h->next_dev++ % nd : 0;
        for (size_t k = 0; k < nd; k++) {
            DevState *d = h->devs[(d0 + k) % nd].get();
            if (on_device >= 0 && d->device != on_device) continue;
            const int pick = d->free_ctxs.empty() ? -1 : (int)d->free_ctxs.size() - 1;  // (the one released last: its buffers are the warmest)
            const bool may_create = d->ctxs.size() < h->max_ctx;
            const bool take = pick >= 0;
            if (!take && !may_create) continue;
            if (!bd || d->in_use < bd->in_use) {
                bd = d;
                bpick = take ? pick : -1;
                bcreate = !take;
            }
        }
        if (bd && !bcreate) {
            c = bd->free_ctxs[bpick];
            bd->free_ctxs.erase(bd->free_ctxs.begin() + bpick);
            bd->in_use++;
            bd->calls++;
            break;
        }
        if (bd) {
            std::unique_ptr<PassCtx> nc;
            rc = new_ctx(h, bd, &nc, (int)bd->ctxs.size());
            if (rc) return rc;
            c = nc.get();
            bd->ctxs.push_back(std::move(nc));
            bd->in_use++;
            bd->calls++;
            break;
        }
        if (on_device >= 0) {
            bool any = false;
            for (auto &d : h->devs) any = any || d->device == on_device;
            if (!any) return fail(ACL_ERR_INVALID_ARGUMENT, "the device buffers live on a device this engine holds no replica on");
        }
        if (opts.cancel || opts.deadline_ns) {
            h->pool_cv.wait_for(lk, std::chrono::microseconds(500));
            rc = check_opts(opts);
            if (rc) return rc;
        } else {
            h->pool_cv.wait(lk);
        }
    }
    lk.unlock();
    // everything this call allocates, copies and launches happens on the context's device
    HIP_TRY(hipSetDevice(c->dev->device));
    c->opts = opts;
    c->timing = h->timing.load(std::memory_order_relaxed);
    return ACL_OK;
}

void Eval::end() {
    if (c) {
        merge_stats(h, c);
        c->opts = CallOpts();
        {
            std::lock_guard<std::mutex> lk(h->pool_mu);
            c->dev->free_ctxs.push_back(c);
            c->dev->in_use--;
        }
        // every waiter: one that asked for a replica on a particular device cannot use a context of another one, and a wake-up it consumed
        // would leave a second waiter asleep next to a free context
        h->pool_cv.notify_all();
        c = nullptr;
    }
    if (locked) {
        h->state_mu.unlock_shared();
        locked = false;
    }
}

constexpr int kTakeLevelLoop = -1000;  // internal: the single-launch path declines the batch (never leaves this file)

// Small batches: the kernel's last block stores `val` into a pinned word behind a system-scope release of everything the launch wrote into host
// memory; the caller spins on that word instead of entering hipStreamSynchronize, which returns ~5.5 us after the store is visible
// (tools/launch_latency.hip).  false: the word did not arrive within 2 ms (a fault, a debugger) -- the caller falls back to the synchronising wait,
// which reports the error.  The stream is left un-synchronised on purpose: everything else on it is ordered behind the kernel anyway.
static bool spin_for(const volatile uint32_t *word, uint32_t val) {
    const int64_t t_end = mono_ns() + 2000000;
    for (uint32_t it = 0;; it++) {
        if (*word == val) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return true;
        }
        if ((it & 1023u) == 1023u && mono_ns() > t_end) return false;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
static uint32_t next_done_val(PassCtx *c) {
    uint32_t v = ++c->done_seq;
    if (!v) v = c->done_seq = 1;  // (0 never names a launch)
    return v;
}

// Small batch: ONE launch (k_check_local) seeds, walks every level and writes the answers.  Requests per wave: one while the
// batch fits the chip's wave slots (latency), more beyond that.  The waves' private frontier regions are carved from
// the context's frontier buffers.
// Geometry of a single-launch pass over n requests: requests per unit, blocks to launch, private frontier entries per block.
struct LocalGeom {
    uint32_t rpw, nblocks, nunits, cap;
    uint32_t nstatic = 0, rdyn = 0;  // != 0: static units for the head of the batch, small hand-out units for its tail
    bool wide = false;               // 12 waves per block and unit (kLocalWide) (chip-filling batches) instead of 4
};
static LocalGeom local_geom(acl_engine *h, PassCtx *c, uint32_t n) {
    LocalGeom G{};
    G.wide = n >= h->local_wide_min;
    const uint32_t blocks = (uint32_t)(G.wide ? c->dev->local_blocks_wide : c->dev->local_blocks);  // what is resident at once; a unit is walked by one block (4 or 12 waves)
    // latency: while the batch has fewer requests than the chip has blocks, every request gets a block of its own; beyond that
    // every block gets ONE unit of n / blocks requests (`upw` > 1: several smaller ones, a second round of per-level chains)
    G.rpw = n <= blocks ? 1u : std::min<uint32_t>(std::max<uint32_t>((n + blocks * h->local_upw - 1) / (blocks * h->local_upw), 1), local_unit_max(G.wide));
    G.nunits = (n + G.rpw - 1) / G.rpw;
    G.nblocks = std::max<uint32_t>(std::min<uint32_t>(G.nunits, blocks), 1);
    // Chip-filling batches: requests differ 100-fold in work, a block's unit is the sum of ~128 of them, and the slowest of 2 048 such sums sets
    // the launch (waves resident 76 % of it, profiles/r02_pmc_walk_final.txt).  So only `local_static_pct` of the batch goes out as one big unit
    // per resident block; the rest is cut into units of `local_dyn_unit` requests that blocks draw from a counter as they finish -- spread over
    // the tail, so the counter's same-address cost (~12 ns per draw) never sees all blocks at once.
    if (h->local_static_pct < 100 && h->local_upw == 1 && G.nunits == blocks && G.rpw >= 2 * h->local_dyn_unit) {
        const uint32_t rs = std::max<uint32_t>(h->local_dyn_unit, (uint32_t)((uint64_t)G.rpw * h->local_static_pct / 100));
        G.rpw = rs;
        G.nstatic = blocks;
        G.rdyn = h->local_dyn_unit;
        G.nunits = G.nstatic + (n - G.nstatic * rs + G.rdyn - 1) / G.rdyn;
    }
    // (a block that needs more than 256 K entries is walking something the whole chip should walk: the level loop takes the batch)
    G.cap = (uint32_t)std::min<uint64_t>(c->frontier_entries / G.nblocks, 1u << 18);
    if (h->local_cap_limit) G.cap = std::min(G.cap, h->local_cap_limit);
    return G;
}

int combine_prepare(acl_engine *h, PassCtx *c, DevGraph *g, uint32_t n, uint32_t blocks, uint32_t rpw) {
    if (!h->snap.has_combine) return ACL_OK;
    // nodes: one per visited state with a combine program; cells: its leaves (<= kMaxLeaves, typically 2-3).  A block of the single-launch walk
    // that outgrows its share sends the batch to the level loop; the level loop's pool running out fails the call.
    uint64_t node_cap, regions = 1;
    if (blocks) {
        node_cap = std::max<uint64_t>(1024, (uint64_t)rpw * 8);
        regions = blocks;
    } else {
        node_cap = std::min<uint64_t>(std::max<uint64_t>((uint64_t)1 << 20, (uint64_t)n * 16), (uint64_t)1 << 26);
    }
    const uint64_t cell_cap = node_cap * 4;
    if ((uint64_t)n + regions * cell_cap >= 0xFFFFFFF0ull) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "combine cells beyond 2^32: lower max_sub_batch");
    HIP_TRY(c->d_nodes.ensure(regions * node_cap));
    HIP_TRY(c->d_has.ensure((size_t)n + regions * cell_cap));
    HIP_TRY(c->d_err.ensure((size_t)n + regions * cell_cap));
    g->bexpr = c->dev->d_bexpr.p;
    g->nodes = c->d_nodes.p;
    g->node_cap = (uint32_t)node_cap;
    g->cell_cap = (uint32_t)cell_cap;
    g->cell0 = n;
    g->ccount = blocks ? nullptr : c->d_status.p + 2 * kLevelSlots + 2;  // (the per-destination export counters of the sharded walk: unused here, reset by k_seed)
    return ACL_OK;
}

// enqueue-only half (memset of the flag, the launch, the flag's read-back)
static int local_enqueue(acl_engine *h, PassCtx *c, const DevGraph &g0, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout) {
    const LocalGeom G = local_geom(h, c, n);
    if (G.cap < 256) return kTakeLevelLoop;
    DevGraph g = g0;
    if (int rc = combine_prepare(h, c, &g, n, G.nblocks, G.rpw)) return rc;
    uint32_t *d_over = c->d_status.p + 2 * kLevelSlots;  // [0] overflow flag, [1] next unit (the sharded walk's export counter: unused here)
    HIP_TRY(hipMemsetAsync(d_over, 0, 3 * sizeof(uint32_t), c->stream));  // [2]: deepest level (a per-destination export counter of the sharded walk: unused here)
    ev_begin(c, 2);
    launch_check_local(c->stream, g, d_items, n, G.rpw, G.nblocks, G.nunits > G.nblocks ? d_over + 1 : nullptr, c->d_fbuf[0].p, c->d_fbuf[1].p, G.cap, d_over, c->d_has.p,
                       c->d_err.p, d_perm, d_errout, d_over + 2, G.nstatic, G.rdyn, G.wide);
    ev_end(c);
    HIP_TRY(hipMemcpyAsync(c->h_status, d_over, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    return ACL_OK;
}
static int local_finish(acl_engine *h, PassCtx *c, uint32_t n) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    ev_collect(c);
    if (c->h_status[0] == 2) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "a relationship row exceeds the per-task enumeration limit");
    if (c->h_status[0] == kOverflowDirect) {
        h->walk_no_direct.store(true, std::memory_order_relaxed);
        c->direct_tripped = true;
    }
    if (c->h_status[0]) return kTakeLevelLoop;
    c->stats.levels_last = c->h_status[2];
    c->stats.check_items += n;
    c->stats.check_passes++;
    c->stats.local_passes++;
    return ACL_OK;
}
int check_pass_local(acl_engine *h, PassCtx *c, const DevGraph &g, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout) {
    const int rc = local_enqueue(h, c, g, d_items, n, d_perm, d_errout);
    return rc ? rc : local_finish(h, c, n);
}

static bool walk_allowed(acl_engine *h, size_t n);
static void walk_outcome(acl_engine *h, PassCtx *c, size_t n, int rc);

// The same for a batch in HOST memory, with no copy engine in the path: the kernel reads the items from pinned host memory
// and writes the answers (and its overflow flag) straight back into pinned host memory, so a pass is ONE launch and ONE
// stream synchronisation -- no H2D, no flag memset, no D2H copies, each of which costs a few microseconds of API time that a
// 64-item batch cannot amortise.  Returns ACL_ERR_RESOURCE_EXHAUSTED (quietly) when the batch must take the level loop.
static int check_pass_local_host(acl_engine *h, PassCtx *c, const acl_item_t *items, uint32_t n, uint8_t *perm_out, int32_t *err_out, bool *attempted) {
    *attempted = false;
    // A batch beyond what one launch takes with one unit per resident block (524 288 items on this chip) goes as SUB-PASSES of equal size: one
    // launch each, back to back on the context's stream, every one reading its slice of the items from -- and answering into -- the caller's
    // pinned memory, ONE synchronisation for all (round 4: this retired the copying pipeline -- look-ahead H2D, lanes, kernels chained on the
    // device through events, a completer thread -- that served only such batches).
    const bool wide0 = n >= h->local_wide_min;
    const uint64_t per_launch = (uint64_t)(wide0 ? c->dev->local_blocks_wide : c->dev->local_blocks) * local_unit_max(wide0);
    uint32_t npass = (uint32_t)std::max<uint64_t>(1, ((uint64_t)n + per_launch - 1) / per_launch);
    // CONCURRENT slices: the items of a launch cross PCIe while its blocks wait for their seeds (4 MB per 262 144 items: ~80 us in which a lone
    // caller's chip idles).  On two streams, the first slice's blocks fill the chip and compute while the second slice's blocks -- which move in
    // as those finish -- fetch theirs; each stream walks in a frontier region of its own.  Sub-passes of a batch beyond one launch always
    // alternate between the two streams (1 048 576 items from one caller: 800 -> 933 M decisions/s); a batch that fits one launch is cut in two
    // only for a LONE caller and only from 262 144 items on (0.336 -> 0.324 ms): the halves' units are half as long, which costs three concurrent
    // callers a tenth of their throughput, and three or four slices lose outright (profiles/r04_host_split.txt).  Not with combine schemas (the
    // slices would share the node / cell scratch) and not while kernels are being timed (the events sit on the context's stream).
    uint32_t nstreams = 1;
    bool lone = false;
    if (n >= 65536) {
        std::lock_guard<std::mutex> lk(h->pool_mu);
        lone = c->dev->in_use <= 1;
    }
    const bool lone_split = lone && npass == 1 && n >= 262144;
    if (h->host_split > 1 && !h->snap.has_combine && !c->timing && (npass > 1 || lone_split)) {
        nstreams = std::min<uint32_t>(h->host_split, 4);
        npass = std::max(npass, nstreams);
    }
    const uint32_t chunk = npass == 1 ? n : (uint32_t)((((uint64_t)n + npass - 1) / npass + 63) / 64 * 64);
    LocalGeom G = local_geom(h, c, std::min(n, chunk));
    G.cap /= nstreams;
    if (G.cap < 256 || n > h->hostmap_max || G.nunits > G.nblocks || npass > 15) return kTakeLevelLoop;
    for (uint32_t k = 1; k < nstreams; k++)
        if (!c->aux[k - 1]) HIP_TRY(hipStreamCreateWithFlags(&c->aux[k - 1], hipStreamNonBlocking));
    *attempted = true;
    HIP_TRY(c->d_has.ensure(std::max<size_t>(n, 4096)));
    HIP_TRY(c->d_err.ensure(std::max<size_t>(n, 4096)));
    // the caller's own buffers where they are pinned (acl_host_alloc), else the context's pinned staging
    const void *src = items;
    if ((const void *)items != c->h_in.p && !h->is_pinned(items, (size_t)n * sizeof(acl_item_t))) {
        HIP_TRY(c->h_in.ensure((size_t)n * sizeof(acl_item_t)));
        std::memcpy(c->h_in.p, items, (size_t)n * sizeof(acl_item_t));
        src = c->h_in.p;
    }
    const bool pin_p = h->is_pinned(perm_out, n), pin_e = err_out && h->is_pinned(err_out, (size_t)n * sizeof(int32_t));
    HIP_TRY(c->h_out.ensure(64 + (size_t)n * 5));
    uint32_t *flag = (uint32_t *)c->h_out.p;  // one overflow flag per sub-pass (16 words)
    int32_t *h_err = pin_e ? err_out : (int32_t *)((char *)c->h_out.p + 64);
    uint8_t *h_perm = pin_p ? perm_out : (uint8_t *)c->h_out.p + 64 + (size_t)n * 4;
    std::memset(flag, 0, 64);  // (word 15: the completion word of small batches)
    const bool spin = npass == 1 && nstreams == 1 && !c->timing && n <= h->spin_max && !h->snap.has_combine;
    const uint32_t done_val = spin ? next_done_val(c) : 0u;
    void *d_in = nullptr, *d_flag = nullptr, *d_perm = nullptr, *d_errp = nullptr;
    // (the staging buffers' device pointers are kept with them; only a caller's own pinned buffer is asked for)
    if (src == c->h_in.p) d_in = c->h_in.dp;
    else HIP_TRY(hipHostGetDevicePointer(&d_in, const_cast<void *>(src), 0));
    d_flag = c->h_out.dp;
    if (pin_p) HIP_TRY(hipHostGetDevicePointer(&d_perm, h_perm, 0));
    else d_perm = (char *)c->h_out.dp + ((char *)h_perm - (char *)c->h_out.p);
    if (pin_e) HIP_TRY(hipHostGetDevicePointer(&d_errp, h_err, 0));
    else d_errp = (char *)c->h_out.dp + ((char *)h_err - (char *)c->h_out.p);
    // (No turn-taking between callers here, whatever the batch size: two single-launch kernels on the chip at once do not get in each other's
    //  way -- the second one's blocks move in as the first one's finish, which fills the tail a lone launch leaves idle: 2 / 4 / 8 / 16 callers
    //  with 262 144-item batches measure 886 / 890 / 914 / 916 M decisions/s, ABOVE the 873 M/s of back-to-back device-resident launches, and a
    //  host mutex around launch + synchronise costs a third of that; profiles/r03_hostmapped_batches.txt.)
    DevGraph g = h->dev_graph(c);
    if (int rc = combine_prepare(h, c, &g, std::min(n, chunk), G.nblocks, G.rpw)) return rc;  // (the sub-passes follow each other on one stream: they share the scratch)
    for (uint32_t k = 0; k < npass; k++) {
        const uint32_t off = k * chunk, m = std::min(chunk, n - off);
        LocalGeom Gk = G;
        if (k + 1 == npass && npass > 1 && m != chunk) {  // (the last one may be shorter)
            Gk = local_geom(h, c, m);
            Gk.cap = std::min(G.cap, Gk.cap);
        }
        const uint32_t lane = k % nstreams;  // slices of one lane follow each other on its stream and share its frontier region
        const size_t region = (size_t)lane * (c->frontier_entries / nstreams);
        hipStream_t st = lane ? c->aux[lane - 1] : c->stream;
        if (nstreams == 1) ev_begin(c, 2);
        launch_check_local(st, g, (const uint4 *)d_in + off, m, Gk.rpw, Gk.nblocks, nullptr, c->d_fbuf[0].p + region, c->d_fbuf[1].p + region, Gk.cap, (uint32_t *)d_flag + k,
                           c->d_has.p + off, c->d_err.p + off, (uint8_t *)d_perm + off, (int32_t *)d_errp + off, nullptr, 0, 0, Gk.wide,
                           lone && Gk.nunits > 1 ? Gk.rpw * h->host_skew_pct / 100 : 0u, spin ? c->d_done.p : nullptr, spin ? (uint32_t *)d_flag + 15 : nullptr, done_val,
                           npass == 1 && n <= 4 ? (const uint4 *)items : nullptr);
        if (nstreams == 1) ev_end(c);
    }
    const bool spun = spin && spin_for(flag + 15, done_val);  // (small batches: see spin_for)
    if (!spun) HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t k = 1; k < nstreams; k++) HIP_TRY(hipStreamSynchronize(c->aux[k - 1]));
    ev_collect(c);
    for (uint32_t k = 0; k < npass; k++)
        if (flag[k] == 2) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "a relationship row exceeds the per-task enumeration limit");
    for (uint32_t k = 0; k < npass; k++)
        if (flag[k] == kOverflowDirect) {
            h->walk_no_direct.store(true, std::memory_order_relaxed);
            c->direct_tripped = true;
        }
    for (uint32_t k = 0; k < npass; k++)
        if (flag[k]) return kTakeLevelLoop;
    if (!pin_p) std::memcpy(perm_out, h_perm, n);
    if (err_out && !pin_e) std::memcpy(err_out, h_err, (size_t)n * sizeof(int32_t));
    c->stats.check_items += n;
    c->stats.check_passes += npass;
    c->stats.local_passes += npass;
    return ACL_OK;
}

// would check_pass_local_host take a batch of n items?  (what the submit pipeline asks before it decides who runs a ticket)
bool hostmap_takes(acl_engine *h, size_t n) {
    if (!(n <= h->local_max_items && n <= h->max_sub_batch && h->shard.world == 1 && n <= h->hostmap_max)) return false;
    const bool wide = n >= h->local_wide_min;
    return n <= 15 * (uint64_t)(wide ? h->dev0().local_blocks_wide : h->dev0().local_blocks) * local_unit_max(wide);  // (up to 15 sub-passes of one unit per resident block; replicas are alike)
}

// A graph whose walks keep outgrowing the blocks' private regions should not pay for a failed walk before every level loop: after an
// overflow the walk sits out 2, 4, ... 64 large passes before it is tried again.
static bool walk_allowed(acl_engine *h, size_t n) {
    if (n < kComputeTokenItems) return true;
    return !(h->local_skip.load(std::memory_order_relaxed) > 0 && h->local_skip.fetch_sub(1, std::memory_order_relaxed) > 0);
}
static void walk_outcome(acl_engine *h, PassCtx *c, size_t n, int rc) {
    // (what THIS call's walk met, kept in its own context: an engine-wide flag let two concurrent callers swap outcomes -- ADVICE r5)
    const bool direct = c->direct_tripped;
    c->direct_tripped = false;
    if (n < kComputeTokenItems) return;
    if (rc == kTakeLevelLoop && direct) return;  // (not a frontier overflow: the next walk simply builds its task lists the general way)
    if (rc == kTakeLevelLoop) h->local_skip.store(1 << std::min(6, 1 + h->local_fail_streak.fetch_add(1, std::memory_order_relaxed)), std::memory_order_relaxed);
    else if (!rc) h->local_fail_streak.store(0, std::memory_order_relaxed);
}

constexpr int kRetryMerging = -1002;  // internal: the level loop ran out of frontier on its first attempt

// the level-synchronous pass (one k_expand launch per dispatch level); `merging`: duplicate entries are struck after every level
static int levels_pass(acl_engine *h, PassCtx *c, const DevGraph &g0, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout, bool merging_asked) {
    // Schemas with `&` / `-`: entries carry result CELLS where the request would be; duplicates are merged on (cell, state, level) through a
    // two-part key (k_dedup_cells), before the duplicate could create a combine node of its own; the combine nodes are evaluated behind the
    // last level, before the answers are written.
    const bool combine = h->snap.has_combine, merging = merging_asked;
    DevGraph g = g0;
    if (int rc = combine_prepare(h, c, &g, n, 0, 0)) return rc;
    for (int attempt = 0;; attempt++) {
        if ((uint64_t)n > c->frontier_entries) {
            int rc = alloc_frontier(h, c, (uint64_t)n * 4);
            if (rc) return rc;
        }
        uint32_t bits = 0;
        if (merging) {
            while ((1ull << bits) < 2 * c->frontier_entries) bits++;
            HIP_TRY(c->d_dedup.ensure(((size_t)1 << bits) + (combine ? (size_t)1 << (bits - 1) : 0)));  // (+ 2^bits u32 second halves)
        }
        DevFrontier f = h->dev_frontier(*c);
        ev_begin(c, 0);
        launch_seed(c->stream, g, f, d_items, n, c->d_has.p, c->d_err.p);  // also resets the status block
        ev_end(c);
        uint32_t levels = 0;
        int rc = level_loop(
            h, c, kMaxLevels,
            [&](uint32_t it) {
                launch_expand(c->stream, g, f, it, c->d_has.p, c->d_err.p);
                if (merging) launch_dedup(c->stream, f, it, c->d_dedup.p, bits, combine);
            },
            &levels,
            [&] {
                if (combine) return;  // (no speculative epilogue: a cell read before its walk is over would turn `a - b` true for good)
                ev_begin(c, 0);
                launch_finalize(c->stream, n, c->d_has.p, c->d_err.p, d_perm, d_errout);
                ev_end(c);
            });
        if (!rc && combine) {
            for (uint32_t it = levels + 1; it >= 1; it--) launch_resolve(c->stream, g, it, c->d_has.p, c->d_err.p);
            launch_finalize(c->stream, n, c->d_has.p, c->d_err.p, d_perm, d_errout);
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        // (combine schemas: a pass that ran out of nodes / cells is first redone with duplicates merged as well -- on a cyclic graph every
        //  repeated visit of a non-monotone state was about to create a node of its own)
        if (rc == ACL_ERR_RESOURCE_EXHAUSTED && combine && !merging_asked && c->h_status[2 * kLevelSlots] == 3) return kRetryMerging;
        if (rc == ACL_ERR_RESOURCE_EXHAUSTED && c->h_status[2 * kLevelSlots] == 1) {
            if (!merging_asked) return kRetryMerging;
            // out of chunks even with duplicates merged: grow (up to 2^28 entries) and redo the pass
            c->stats.overflow_retries++;
            if (c->frontier_entries >= (uint64_t)kMaxFrontierChunks * kChunk || attempt > 8)
                return fail(ACL_ERR_RESOURCE_EXHAUSTED, "frontier capacity exceeded (" + std::to_string(c->frontier_entries) + " entries); lower max_sub_batch");
            int rc2 = alloc_frontier(h, c, c->frontier_entries * 4);
            if (rc2) return rc2;
            continue;
        }
        if (rc) return rc;
        c->levels_hint = levels;
        c->stats.levels_last = levels;
        c->stats.check_items += n;
        c->stats.check_passes++;
        return ACL_OK;
    }
}

// one device pass over n (<= max_sub_batch) interned items already in HBM
int check_pass(acl_engine *h, PassCtx *c, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout, bool try_local) {
    HIP_TRY(c->d_has.ensure(std::max<size_t>(n, 4096)));
    HIP_TRY(c->d_err.ensure(std::max<size_t>(n, 4096)));
    DevGraph g = h->dev_graph(c);
    // small batches (the proxy's own call shape: check.go:76-94, watch.go:50): ONE launch runs every level, each wave
    // walking its own slice of the batch through a wave-private frontier -- no host round trip between levels
    if (try_local && n <= h->local_max_items && walk_allowed(h, n)) {
        int rc = check_pass_local(h, c, g, d_items, n, d_perm, d_errout);
        walk_outcome(h, c, n, rc);
        if (rc != kTakeLevelLoop) return rc;  // kTakeLevelLoop: a block ran out of private frontier, the level-synchronous path takes the batch
    }
    int rc = levels_pass(h, c, g, d_items, n, d_perm, d_errout, false);
    if (rc != kRetryMerging) return rc;
    // The frontier outgrew its buffers.  Before growing them: merge identical (request, state, level) entries after every level
    // (k_dedup) -- nested groups with branching cycles double the frontier per level otherwise -- on slices the dedup key can hold.
    c->stats.overflow_retries++;
    for (uint32_t off = 0; off < n; off += kDedupBatch) {
        const uint32_t m = std::min<uint32_t>(kDedupBatch, n - off);
        rc = levels_pass(h, c, g, d_items + off, m, d_perm + off, d_errout ? d_errout + off : nullptr, true);
        if (rc) return rc;
    }
    return ACL_OK;
}

int not_sharded(acl_engine *h) {
    if (h->shard.world > 1)
        return fail(ACL_ERR_FAILED_PRECONDITION, "this engine holds one shard of the graph: evaluate through acl_shard_* with the other shards");
    return ACL_OK;
}

int check_device(acl_engine *h, PassCtx *c, const uint4 *d_items, size_t n, uint8_t *d_perm, int32_t *d_errout, bool try_local) {
    int rc = not_sharded(h);
    if (rc) return rc;
    for (size_t b = 0; b < n; b += h->max_sub_batch) {
        uint32_t m = (uint32_t)std::min<size_t>(h->max_sub_batch, n - b);
        rc = check_pass(h, c, d_items + b, m, d_perm + b, d_errout ? d_errout + b : nullptr, try_local);
        if (rc) return rc;
    }
    return ACL_OK;
}

// Host items in, host answers out (SURVEY.md 8(d) timing variant (ii)): H2D, kernels, D2H on the context's stream.
// Buffers from acl_host_alloc are pinned and are DMA'd directly; anything else is staged through the context's pinned
// buffers (an async copy from pageable memory would be staged by the runtime anyway, synchronously).
int check_ids_host(acl_engine *h, PassCtx *c, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, bool items_on_device) {
    const bool walk = n <= h->local_max_items && n <= h->max_sub_batch && h->shard.world == 1;
    const bool allowed = walk && walk_allowed(h, n);  // (asked once per call: it counts down the back-off after an overflow)
    bool tried = false;  // the single-launch walk has had its go at this batch
    if (allowed && !items_on_device) {
        // first choice at every size: the kernel reads the items from, and writes the answers to, pinned host memory itself -- one launch, one
        // synchronisation, no copy engine, no turn-taking between callers
        int rc = check_pass_local_host(h, c, items, (uint32_t)n, perm_out, err_out, &tried);
        if (tried) walk_outcome(h, c, n, rc);
        if (rc != kTakeLevelLoop) return rc;
        // a block ran out of private frontier (-> the level loop below), or the batch needs more units than blocks (-> the copying walk below)
    }
    if (!items_on_device) HIP_TRY(c->d_items.ensure(n));  // (a grow-only buffer whose contents an ensure() may discard)
    HIP_TRY(c->d_perm.ensure(n));
    HIP_TRY(c->d_errout.ensure(n));
    if (!items_on_device) {
        const void *src = items;
        if ((const void *)items != c->h_in.p && !h->is_pinned(items, n * sizeof(acl_item_t))) {  // (the string entry points intern straight into the staging buffer)
            HIP_TRY(c->h_in.ensure(n * sizeof(acl_item_t)));
            std::memcpy(c->h_in.p, items, n * sizeof(acl_item_t));
            src = c->h_in.p;
        }
        HIP_TRY(hipMemcpyAsync(c->d_items.p, src, n * sizeof(acl_item_t), hipMemcpyHostToDevice, c->stream));
    }
    const bool pin_p = h->is_pinned(perm_out, n), pin_e = !err_out || h->is_pinned(err_out, n * sizeof(int32_t));
    uint8_t *hp = perm_out;
    int32_t *he = err_out;
    if (!pin_p || !pin_e) {
        HIP_TRY(c->h_out.ensure(n * 5 + 64));
        if (!pin_e) he = (int32_t *)c->h_out.p;
        if (!pin_p) hp = (uint8_t *)c->h_out.p + n * 4;
    }
    auto results_d2h = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(hp, c->d_perm.p, n, hipMemcpyDeviceToHost, c->stream));
        if (err_out) HIP_TRY(hipMemcpyAsync(he, c->d_errout.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        return ACL_OK;
    };
    // (What is left of the copying path: batches the host-mapped walk above does not take -- switched off, backing off after an overflow, beyond
    // its sub-pass limit.)  ONE launch where the single-launch walk takes it: H2D, kernel, overflow-flag read-back and result copies all go onto
    // the stream and the host synchronises ONCE.  Only a walk that overflowed its private regions comes back for the level loop.
    int rc = kTakeLevelLoop;
    if (allowed && !tried) {
        tried = true;
        HIP_TRY(c->d_has.ensure(std::max<size_t>(n, 4096)));
        HIP_TRY(c->d_err.ensure(std::max<size_t>(n, 4096)));
        // chip-filling batches of several callers: kernels one at a time (host mutex, held from the launch to the one synchronisation); the
        // next caller's H2D, already enqueued on its own stream, runs meanwhile
        std::unique_lock<std::mutex> tk(c->dev->compute_mu, std::defer_lock);
        if (n >= kComputeTokenItems) tk.lock();
        rc = local_enqueue(h, c, h->dev_graph(c), c->d_items.p, (uint32_t)n, c->d_perm.p, c->d_errout.p);
        if (!rc) {
            rc = results_d2h();
            if (rc) return rc;
            rc = local_finish(h, c, (uint32_t)n);
        } else if (rc != kTakeLevelLoop) {
            (void)hipStreamSynchronize(c->stream);
        }
        walk_outcome(h, c, n, rc);
    }
    if (rc == kTakeLevelLoop) {  // a block ran out of private frontier, or the walk is switched off / backing off: the level loop, one batch at a time
        HIP_TRY(hipStreamSynchronize(c->stream));
        {
            std::unique_lock<std::mutex> tk(c->dev->compute_mu, std::defer_lock);
            if (n >= kComputeTokenItems) tk.lock();
            rc = check_device(h, c, c->d_items.p, n, c->d_perm.p, c->d_errout.p, !tried);  // (ends with the context's stream synchronised; a batch the walk has not tried -- sub-batched ones -- tries it per pass)
        }
        if (rc) return rc;
        rc = results_d2h();
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        ev_collect(c);
    }
    if (rc) return rc;
    if (!pin_p) std::memcpy(perm_out, hp, n);
    if (err_out && !pin_e) std::memcpy(err_out, he, n * sizeof(int32_t));
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

int32_t intern_check_item(acl_engine_t *h, const acl_check_item_t &it, acl_item_t *out) {
    const Schema &sc = h->store.schema();
    if (empty(it.resource_type) || empty(it.resource_id) || empty(it.permission) || empty(it.subject_type) || empty(it.subject_id))
        return ACL_ERR_INVALID_ARGUMENT;  // empty request: pkg/proxy/options_test.go:101-102
    int rt = sc.type_of(it.resource_type), st = sc.type_of(it.subject_type);
    int pm = rt < 0 ? -1 : sc.defs[rt].find(it.permission);
    int sr = kNoRelation;
    bool bad = rt < 0 || st < 0 || pm < 0;
    const bool srel_given = !empty(it.subject_relation) && std::strcmp(it.subject_relation, "...") != 0;
    if (srel_given) {
        sr = st < 0 ? -1 : sc.defs[st].find(it.subject_relation);
        bad = bad || sr < 0;
    }
    // API validation beats "not found" (validate.hpp): ill-formed names and ids, and `*` anywhere in a Check
    if ((rt < 0 && !valid_type_name(it.resource_type)) || (st < 0 && !valid_type_name(it.subject_type)) || (pm < 0 && !valid_relation_name(it.permission)) ||
        (srel_given && sr < 0 && !valid_relation_name(it.subject_relation)))
        return ACL_ERR_INVALID_ARGUMENT;
    if (bad) {
        if (!valid_object_id(it.resource_id) || !valid_object_id(it.subject_id)) return ACL_ERR_INVALID_ARGUMENT;
        return ACL_ERR_FAILED_PRECONDITION;
    }
    // unknown object ids have no relationships: sentinels above every dense id, equal only when
    // resource and subject are the same (unknown) object
    uint32_t res, sub;
    bool kr = h->store.objects(rt).find(it.resource_id, &res), ks = h->store.objects(st).find(it.subject_id, &sub);
    // (these ids leave the names lock in the caller's hands -- a single Check queued in the batcher: their recycling quarantine starts over, store.hpp touch)
    if (kr) h->store.touch(rt, res);
    if (ks) h->store.touch(st, sub);
    // (every name IN a table passed the id pattern when it was interned -- except "*", the wildcard subject's name: only unknown ids are spelled out)
    if ((!kr && !valid_object_id(it.resource_id)) || (!ks && !valid_object_id(it.subject_id)) || std::strcmp(it.resource_id, "*") == 0 || std::strcmp(it.subject_id, "*") == 0)
        return ACL_ERR_INVALID_ARGUMENT;
    if (!kr && !ks && rt == st && std::strcmp(it.resource_id, it.subject_id) == 0) res = sub = 0xFFFFFFFEu;
    else {
        if (!kr) res = 0xFFFFFFFDu;
        if (!ks) sub = 0xFFFFFFFCu;
    }
    *out = acl_item_t{(uint16_t)rt, (uint16_t)pm, res, (uint16_t)st, (uint16_t)(sr == kNoRelation ? ACL_NO_RELATION : sr), sub};
    return 0;
}

// The string entry points' host half (SURVEY.md 7 "the GPU is not the bottleneck; the host is").  Two item forms share one core:
// NUL-terminated fields (acl_check_item_t) and {pointer, length} fields (acl_check_item_v_t -- what a cgo shim can point at Go string
// data without copying).  Type / permission names repeat across a bulk request (check.go:23-39 resolves one rule template per item), so
// the last resolved (type, permission, subject type, subject relation) is remembered per thread and recognised BY POINTER first: the
// same template hands over the same string.  Object ids: hash, then the table's three-stage pipelined lookup over groups of items.
struct CStrItems {
    const acl_check_item_t *it;
    static constexpr bool kHasLen = false;
    const char *ptr(size_t i, int f) const { return (&it[i].resource_type)[f]; }
    size_t len(size_t i, int f) const {
        const char *p = ptr(i, f);
        return p ? std::strlen(p) : 0;
    }
};
struct ViewItems {
    const acl_check_item_v_t *it;
    static constexpr bool kHasLen = true;
    const char *ptr(size_t i, int f) const { return (&it[i].resource_type)[f].p; }
    size_t len(size_t i, int f) const { return (&it[i].resource_type)[f].p ? (&it[i].resource_type)[f].n : 0; }
};
// ... and the PACKED form (acl_check_bulk_packed, round 6): a dictionary of the call's DISTINCT strings and six u32 indices per item.  A shim that walks a kube
// list copies every string once anyway (shim/go/aclgpu/engine.go); written into one buffer, an item is 24 bytes instead of six views (96), the constant
// fields of a PostFilter call -- type, permission, the user -- are the SAME dictionary entry (found equal by index, no bytes compared), and a name that
// occurs in many items of the call is resolved once (PackedCache below).
struct PackedItems {
    const acl_packed_request_t *rq;
    static constexpr bool kHasLen = true;
    uint32_t idx(size_t i, int f) const { return rq->items[6 * i + f]; }
    const char *ptr(size_t i, int f) const {
        const uint32_t k = idx(i, f);
        return k == ACL_PACKED_NONE ? nullptr : rq->bytes + rq->offsets[k];
    }
    size_t len(size_t i, int f) const {
        const uint32_t k = idx(i, f);
        return k == ACL_PACKED_NONE ? 0 : rq->offsets[k + 1] - rq->offsets[k];
    }
};
enum { F_RT = 0, F_RID = 1, F_PM = 2, F_ST = 3, F_SID = 4, F_SR = 5 };
// Which field of item i fails the API's validation, and why -- for acl_last_error() ("check failed" told an operator nothing; the
// reference denies everything a failed CheckBulkPermissions asked, pkg/authz/check.go:48-52, so the message is all there is to diagnose a blanket denial).
template <class Items>
static std::string describe_invalid(const Schema &sc, const Items &its, size_t i) {
    static const char *const kField[6] = {"resource type", "resource id", "permission", "subject type", "subject id", "subject relation"};
    auto view = [&](int f) {
        const char *q = its.ptr(i, f);
        return q ? std::string_view(q, its.len(i, f)) : std::string_view();
    };
    auto shown = [](std::string_view v) {
        std::string o(v.substr(0, 48));
        for (char &ch : o)
            if ((unsigned char)ch < 0x20 || (unsigned char)ch > 0x7E) ch = '?';
        return "`" + o + (v.size() > 48 ? "...` (" + std::to_string(v.size()) + " bytes)" : "`");
    };
    const std::string_view rt = view(F_RT), rid = view(F_RID), pm = view(F_PM), st = view(F_ST), sid = view(F_SID);
    std::string_view sr = view(F_SR);
    if (sr == "...") sr = std::string_view();
    const int rti = sc.type_of(std::string(rt)), sti = sc.type_of(std::string(st));
    auto bad_id = [&](int f, std::string_view v) -> std::string {
        if (v.empty()) return std::string(kField[f]) + " is empty";
        if (v == "*") return std::string(kField[f]) + " `*`: a wildcard is not an object of a Check";
        if (v.size() > 1024) return std::string(kField[f]) + " " + shown(v) + " is longer than 1024 bytes";
        size_t at = 0;
        while (at < v.size() && valid_object_id(v.substr(at, 1))) at++;
        if (at < v.size()) return std::string(kField[f]) + " " + shown(v) + " does not match ^[a-zA-Z0-9/_|\\-=+]{1,1024}$ (byte " + std::to_string(at) + " `" +
                                  ((unsigned char)v[at] >= 0x20 && (unsigned char)v[at] <= 0x7E ? std::string(1, v[at]) : std::string("?")) + "`)";
        return std::string();
    };
    if (rt.empty()) return "resource type is empty";
    if (rti < 0 && !valid_type_name(rt)) return "resource type " + shown(rt) + " does not match ^([a-z][a-z0-9_]{1,61}[a-z0-9]/)*[a-z][a-z0-9_]{1,62}[a-z0-9]$";
    if (std::string e = bad_id(F_RID, rid); !e.empty()) return e;
    if (pm.empty()) return "permission is empty";
    if ((rti < 0 || sc.defs[rti].find(std::string(pm)) < 0) && !valid_relation_name(pm)) return "permission " + shown(pm) + " does not match ^[a-z][a-z0-9_]{1,62}[a-z0-9]$";
    if (st.empty()) return "subject type is empty";
    if (sti < 0 && !valid_type_name(st)) return "subject type " + shown(st) + " does not match ^([a-z][a-z0-9_]{1,61}[a-z0-9]/)*[a-z][a-z0-9_]{1,62}[a-z0-9]$";
    if (std::string e = bad_id(F_SID, sid); !e.empty()) return e;
    if (!sr.empty() && (sti < 0 || sc.defs[sti].find(std::string(sr)) < 0) && !valid_relation_name(sr))
        return "subject relation " + shown(sr) + " does not match ^[a-z][a-z0-9_]{1,62}[a-z0-9]$ (or empty, or `...`)";
    return "a field is empty or ill-formed";
}
template <class Items>
static int fail_invalid_item(acl_engine_t *h, const Items &its, size_t i) {
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    return fail(ACL_ERR_INVALID_ARGUMENT, "invalid CheckBulkPermissionsRequest: item " + std::to_string(i) + ": " + describe_invalid(h->store.schema(), its, i));
}
static_assert(offsetof(acl_check_item_t, subject_relation) == 5 * sizeof(const char *), "acl_check_item_t: six consecutive pointers");
static_assert(offsetof(acl_check_item_v_t, subject_relation) == 5 * sizeof(acl_str_t), "acl_check_item_v_t: six consecutive views");

struct NameMemo {
    const char *p[4] = {nullptr, nullptr, nullptr, nullptr};  // resource type, permission, subject type, subject relation: as last seen
    size_t n[4] = {0, 0, 0, 0};
    std::string s[4];
    int rti = -1, pmi = -1, sti = -1, sri = kNoRelation;
    bool bad = true, valid = false;
    bool malformed = false;  // an undeclared name that does not even match the API's pattern: InvalidArgument, not "not found" (validate.hpp)
};

// names -> indices of item i (memoised per thread); false: *err says why the item cannot be checked
template <class Items>
static bool intern_names(const Schema &sc, const Items &its, size_t i, NameMemo &m, int32_t *err) {
    static const int kF[4] = {F_RT, F_PM, F_ST, F_SR};
    bool same = m.valid;
    for (int k = 0; k < 4 && same; k++) same = its.ptr(i, kF[k]) == m.p[k] && (!Items::kHasLen || its.len(i, kF[k]) == m.n[k]);
    if (!same) {
        std::string_view v[4];
        for (int k = 0; k < 4; k++) {
            const char *q = its.ptr(i, kF[k]);
            v[k] = q ? std::string_view(q, its.len(i, kF[k])) : std::string_view();
        }
        if (v[3] == "...") v[3] = std::string_view();
        const bool content = m.valid && v[0] == m.s[0] && v[1] == m.s[1] && v[2] == m.s[2] && v[3] == m.s[3];
        for (int k = 0; k < 4; k++) {
            m.p[k] = its.ptr(i, kF[k]);
            m.n[k] = Items::kHasLen ? its.len(i, kF[k]) : 0;
        }
        if (!content) {
            for (int k = 0; k < 4; k++) m.s[k].assign(v[k].data() ? v[k].data() : "", v[k].size());
            m.rti = sc.type_of(m.s[0]);
            m.sti = sc.type_of(m.s[2]);
            m.pmi = m.rti < 0 ? -1 : sc.defs[m.rti].find(m.s[1]);
            m.sri = kNoRelation;
            m.bad = m.rti < 0 || m.sti < 0 || m.pmi < 0;
            if (!m.s[3].empty()) {
                m.sri = m.sti < 0 ? -1 : sc.defs[m.sti].find(m.s[3]);
                m.bad = m.bad || m.sri < 0;
            }
            m.malformed = (m.rti < 0 && !valid_type_name(m.s[0])) || (m.sti < 0 && !valid_type_name(m.s[2])) || (m.pmi < 0 && !valid_relation_name(m.s[1])) ||
                          (!m.s[3].empty() && m.sri < 0 && !valid_relation_name(m.s[3]));
        }
        m.valid = true;
    }
    // empty request fields: pkg/proxy/options_test.go:101-102 (the subject relation may be empty)
    if (m.s[0].empty() || m.s[1].empty() || m.s[2].empty() || m.malformed) {
        *err = ACL_ERR_INVALID_ARGUMENT;
        return false;
    }
    if (m.bad) {
        *err = ACL_ERR_FAILED_PRECONDITION;
        return false;
    }
    return true;
}

// Host threads of the string entry points' interning: persistent (spawning 15 threads costs 0.2-2 ms per call -- more than interning a
// 64 k-item batch), woken per batch; the caller works too.
struct InternPool {
    // A batch is OPEN between run()'s two stores to `open`.  A worker enters one by counting itself in (`inside`) and THEN reading `open`; run() closes the batch
    // and THEN waits for `inside` to drain: whichever of the two sequentially consistent pairs comes first, either the worker sees the batch closed and leaves
    // without touching it, or run() sees the worker and waits -- `fn` and the batch's fields are never read after run() returned.  No mutex on this path:
    // 31 workers signing in and out of every batch through one lock cost a 65 536-item call 40-60 us per batch, three batches per call (round 6).  Only a worker
    // that has polled kSpinNs for nothing sleeps, on `mu` / `cv`; one that wakes up late finds its batch closed and does not hold anybody up.
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> threads;
    const std::function<void(size_t, size_t)> *job = nullptr;
    size_t n = 0, chunk = 1;
    std::atomic<size_t> next{0};
    unsigned limit = 0;  // workers that take chunks of the current batch
    std::atomic<uint64_t> gen_a{0};
    std::atomic<bool> open{false}, stop_a{false};
    std::atomic<int> inside{0};
    std::atomic<unsigned> sleepers{0};
    static constexpr int64_t kSpinNs = 150000;
    std::mutex call_mu;  // one batch at a time

    // most workers are still polling (a batch ended less than kSpinNs ago): a batch of a few hundred items is worth spreading, nobody has to be woken up
    bool awake() const { return (size_t)sleepers.load(std::memory_order_relaxed) * 2 < threads.size(); }
    // Which piece goes to whom: participant p (the workers 0 .. limit - 1, the caller = limit) takes the pieces p, p + P, p + 2 P, ... first and only then whatever
    // is left (a participant that shows up late loses its pieces to the others).  Two batches over the same items -- the PostFilter route's pass and its test --
    // then meet the same thread per piece: what the first wrote about an item (its hash, its id) is in the cache of the thread that reads it in the second,
    // not a modified line in another core's (12-25 ns per item to pull over, against 1-2).
    std::unique_ptr<std::atomic<uint8_t>[]> taken;
    size_t taken_cap = 0, npieces = 0;
    void work(unsigned me) {
        const size_t P = (size_t)limit + 1;
        auto take = [&](size_t c) {
            if (taken[c].load(std::memory_order_relaxed) || taken[c].exchange(1, std::memory_order_relaxed)) return;
            (*job)(c * chunk, std::min(n, (c + 1) * chunk));
        };
        for (size_t c = me; c < npieces; c += P) take(c);
        for (size_t k = 0, c = me < npieces ? me : 0; k < npieces; k++, c = c + 1 == npieces ? 0 : c + 1) take(c);
    }
    void loop(unsigned me) {
        uint64_t seen = 0;
        for (;;) {
            // A sleep + wake-up costs a thread 20-100 us on these hosts, about what its share of a 16 384-item batch takes: a worker that has just
            // finished a batch polls for the next one for kSpinNs before it goes to sleep (a busy proxy's bulk calls follow each other closely).
            bool got = false;
            for (const auto t0 = std::chrono::steady_clock::now(); seen && !got && std::chrono::steady_clock::now() - t0 < std::chrono::nanoseconds(kSpinNs);) {
                for (int i = 0; i < 64 && !got; i++) {
                    got = gen_a.load(std::memory_order_acquire) != seen || stop_a.load(std::memory_order_relaxed);
                    if (!got) __builtin_ia32_pause();
                }
            }
            if (!got) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    sleepers.fetch_add(1);  // (before the predicate's first look at gen_a: run() bumps gen_a and then reads `sleepers`)
                    cv.wait(lk, [&] { return stop_a.load() || gen_a.load() != seen; });
                    sleepers.fetch_sub(1);
                }
                // the wake-ups fan out: run() wakes two sleepers, each of them two more -- 31 futex wake-ups in a row kept the CALLER from its own share of the
                // batch for 40 us (round 6: "first piece began at 39 us" with every worker asleep)
                if (!stop_a.load() && open.load() && sleepers.load() != 0) {
                    cv.notify_one();
                    cv.notify_one();
                }
            }
            if (stop_a.load()) return;
            seen = gen_a.load(std::memory_order_acquire);
            inside.fetch_add(1);
            if (open.load() && me < limit) work(me);
            inside.fetch_sub(1);
        }
    }
    // The workers stay on the NUMA node of the thread that creates the pool (the first large string batch's caller): the name tables were
    // filled from that side, and on a two-socket host a worker that lands on the other socket pays a remote access for every slot it probes --
    // the same binary measured 0.34 ms or 0.55 ms per 65 536-item call depending on where the scheduler had put the threads
    // (profiles/r03_string_path_ab.txt).  ACL_INTERN_PIN=0: leave them to the scheduler.
    static bool node_cpus(cpu_set_t *out) {
        const int cpu = sched_getcpu();
        if (cpu < 0) return false;
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
        for (int node = 0; node < 64; node++) {
            char path[96];
            std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
            FILE *f = std::fopen(path, "r");
            if (!f) break;
            char buf[4096];
            const bool got = std::fgets(buf, sizeof(buf), f) != nullptr;
            std::fclose(f);
            if (!got) continue;
            CPU_ZERO(out);
            bool mine = false;
            int n = 0;
            for (const char *q = buf; *q && *q != '\n';) {  // "0-63,128-191"
                char *end = nullptr;
                const long a = std::strtol(q, &end, 10);
                if (end == q) break;
                long b = a;
                q = end;
                if (*q == '-') {
                    b = std::strtol(q + 1, &end, 10);
                    q = end;
                }
                for (long c = a; c <= b && c < CPU_SETSIZE; c++)
                    if (CPU_ISSET((int)c, &allowed)) {
                        CPU_SET((int)c, out);
                        n++;
                        mine = mine || c == cpu;
                    }
                if (*q == ',') q++;
            }
            if (mine && n >= 2) return true;
        }
        return false;
    }
    explicit InternPool(unsigned nthreads) {
        for (unsigned i = 0; i < nthreads; i++) threads.emplace_back([this, i] { loop(i); });
        const char *ev = getenv("ACL_INTERN_PIN");
        cpu_set_t set;
        if (!(ev && atoi(ev) == 0) && node_cpus(&set))
            for (auto &t : threads) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
    }
    ~InternPool() {
        stop_a.store(true);
        {
            std::lock_guard<std::mutex> lk(mu);  // (a worker between its predicate and its wait holds mu: the notify below cannot slip in there)
        }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
    // meanwhile: what the CALLER does between starting the batch and joining it (a device call it waits for while the workers go through the items).  It must
    // not take state_mu or names_mu: interning callers wait for call_mu under names_mu (lock order: state_mu, names_mu, call_mu).
    void run(size_t total, size_t chunk_items, unsigned workers, const std::function<void(size_t, size_t)> &fn, const std::function<void()> *meanwhile = nullptr) {
        std::lock_guard<std::mutex> one(call_mu);
        job = &fn;
        n = total;
        chunk = chunk_items;
        limit = workers;
        next.store(0, std::memory_order_relaxed);
        npieces = (total + chunk_items - 1) / chunk_items;
        if (taken_cap < npieces) {
            taken_cap = std::max<size_t>(256, npieces * 2);
            taken.reset(new std::atomic<uint8_t>[taken_cap]);
        }
        for (size_t c = 0; c < npieces; c++) taken[c].store(0, std::memory_order_relaxed);
        open.store(true);
        gen_a.fetch_add(1);
        if (sleepers.load() != 0) {
            {
                std::lock_guard<std::mutex> lk(mu);
            }
            cv.notify_one();
            cv.notify_one();
        }
        if (meanwhile) (*meanwhile)();
        work(limit);
        open.store(false);
        for (unsigned spins = 0; inside.load() != 0; spins++) {  // (workers still in their last chunk)
            if (spins < 4096) __builtin_ia32_pause();
            else std::this_thread::yield();
        }
    }
};

void intern_pool_destroy(acl_engine_t *h) {
    delete h->intern_pool;
    h->intern_pool = nullptr;
}

// fn over [0, total) in pieces, on the interning pool's threads and the caller (engine_list.cpp: a list response's bytes).  Lock order as for the interning
// callers: names_mu (shared) may be held, state_mu must not be waited for inside fn.
void host_parallel(acl_engine_t *h, size_t total, size_t piece, const std::function<void(size_t, size_t)> &fn) {
    piece = std::max<size_t>(piece, 1);
    const unsigned threads = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), h->intern_threads);
    if (threads <= 1 || total <= piece) {
        if (total) fn(0, total);
        return;
    }
    InternPool *P;
    {
        std::lock_guard<std::mutex> lk(h->intern_pool_mu);
        if (!h->intern_pool) h->intern_pool = new InternPool(std::min<unsigned>(std::max(2u, std::thread::hardware_concurrency()), h->intern_threads) - 1);
        P = h->intern_pool;
    }
    P->run(total, piece, h->intern_threads - 1, fn);
}
unsigned host_threads(acl_engine_t *h) { return std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), h->intern_threads); }

// Interns n items into `out` (the evaluation context's pinned staging: what the H2D copy reads).  An item that cannot be checked -- empty
// field, unknown type / permission / relation: the pair carries an error, check.go:55 -- becomes a DEAD item (the kernel answers it
// "invalid" without touching the graph) and is listed in *bad with its error; the batch is never compacted or copied again.
constexpr uint16_t kDeadType = 0xFFFFu;
template <class Items>
static void intern_items(acl_engine_t *h, const Items &its, size_t n, acl_item_t *out, std::vector<std::pair<uint32_t, int32_t>> *bad, bool ids_leave_the_call = false) {
    const Schema &sc = h->store.schema();
    std::mutex bad_mu;
    // Packed requests: a name that many items of the call carry -- the namespace of every pod of a list, the user of every pair -- is looked up ONCE: entry d of
    // this table remembers what dictionary string d resolved to as an object of one type ((type + 1) << 33 | known << 32 | id; 0 = not yet; racing threads store the
    // same value).  Items that name an object for the first time pay the table's DRAM miss as before.
    constexpr bool kPacked = std::is_same_v<Items, PackedItems>;
    std::unique_ptr<std::atomic<uint64_t>[]> dict_cache;
    if constexpr (kPacked) {
        // (only where the dictionary says names REPEAT: a call of K distinct resources and one user has about one string per item, and a table that is
        //  zeroed, filled and never hit cost the 65 536-item call 60 us)
        if (n >= 64 && (size_t)its.rq->n_strings * 2 <= n) {
            dict_cache.reset(new std::atomic<uint64_t>[its.rq->n_strings]);
            for (uint32_t d = 0; d < its.rq->n_strings; d++) dict_cache[d].store(0, std::memory_order_relaxed);
        }
    }
    auto cached = [&](size_t i, int f, int type, bool *known, uint32_t *id) -> bool {
        if constexpr (kPacked) {
            if (!dict_cache) return false;
            const uint64_t v = dict_cache[its.idx(i, f)].load(std::memory_order_relaxed);
            if ((v >> 33) != (uint64_t)type + 1u) return false;
            *known = (v >> 32) & 1u;
            *id = (uint32_t)v;
            return true;
        } else {
            (void)i, (void)f, (void)type, (void)known, (void)id;
            return false;
        }
    };
    auto remember = [&](size_t i, int f, int type, bool known, uint32_t id) {
        if constexpr (kPacked) {
            if (dict_cache) dict_cache[its.idx(i, f)].store(((uint64_t)type + 1u) << 33 | (uint64_t)known << 32 | id, std::memory_order_relaxed);
        } else {
            (void)i, (void)f, (void)type, (void)known, (void)id;
        }
    };
    // Object ids: two lookups per item in tables of up to millions of names -- two dependent DRAM misses each (slot, then the name's
    // bytes).  Items go in groups of kGroup through three stages: hash + prefetch the slots; walk to the tag match + prefetch the names;
    // compare.  The misses of a group are in flight together.
    constexpr size_t kGroup = 16;  // (32: no better on the GPU box's host; prefetching the NEXT group's id bytes ahead of their hashing: 0.33 against 0.32 ms per 65 536 items, not kept -- tools/intern_bench.py)
    static const bool kPieces = getenv("ACL_DEBUG_INTERN_PIECES") != nullptr;
    const int64_t tp0 = kPieces ? mono_ns() : 0;
    std::atomic<int64_t> p_sum{0}, p_first{INT64_MAX}, p_last{0}, p_max{0};
    std::atomic<int> p_threads{0};
    const std::function<void(size_t, size_t)> run = [&](size_t a, size_t b) {
        struct PieceTrace {
            int64_t t0, c0;
            std::atomic<int64_t> *sum, *first, *last, *mx;
            ~PieceTrace() {
                if (!sum) return;
                const int64_t now = mono_ns();
                sum->fetch_add(now - c0);
                for (int64_t v = first->load(); c0 - t0 < v && !first->compare_exchange_weak(v, c0 - t0);) {}
                for (int64_t v = last->load(); now - t0 > v && !last->compare_exchange_weak(v, now - t0);) {}
                for (int64_t v = mx->load(); now - c0 > v && !mx->compare_exchange_weak(v, now - c0);) {}
            }
        } pt{tp0, kPieces ? mono_ns() : 0, kPieces ? &p_sum : nullptr, &p_first, &p_last, &p_max};
        static thread_local int64_t seen_call = 0;
        if (kPieces && seen_call != tp0) { seen_call = tp0; p_threads.fetch_add(1); }
        NameMemo m;
        std::vector<std::pair<uint32_t, int32_t>> mybad;
        const int64_t touch_ms = ids_leave_the_call ? Store::steady_now_ms() : 0;  // (one clock read per chunk, not per id)
        struct Pending {
            uint64_t hr, hs;
            std::string_view rid, sid;
            int rt, st, pm, sr;
            bool ok;
            bool same_res, same_sub;  // the same object as the item before it: its id is taken over, not looked up again
            bool hit_res, hit_sub;    // packed requests: the dictionary entry was resolved earlier in this call (kr / res, ks / sub already hold the answer)
            bool kr, ks;              // (third stage) the table knows the name
            uint32_t res, sub;
        } pend[kGroup];
        // The proxy's batches repeat themselves: every pair of a PostFilter call names the requesting user (postfilter.go:88-119), the F templates
        // of a list item name the same object one after the other, check.go:17-72 builds all of a request's pairs for one user.  A name equal
        // to the previous item's (same type; pointer + length, or content) costs no hash, no prefetch and no probe.
        Pending last{};  // the last item of the previous group that resolved
        bool have_last = false;
        auto same_name = [](std::string_view x, std::string_view y) { return x.size() == y.size() && (x.data() == y.data() || std::memcmp(x.data(), y.data(), x.size()) == 0); };
        for (size_t g0 = a; g0 < b; g0 += kGroup) {
            const size_t g1 = std::min(b, g0 + kGroup);
            for (size_t i = g0; i < g1; i++) {
                Pending &p = pend[i - g0];
                int32_t err = 0;
                const char *r = its.ptr(i, F_RID), *u = its.ptr(i, F_SID);
                p.rid = r ? std::string_view(r, its.len(i, F_RID)) : std::string_view();
                p.sid = u ? std::string_view(u, its.len(i, F_SID)) : std::string_view();
                p.ok = intern_names(sc, its, i, m, &err);
                // an empty or ill-formed id beats an unknown name (API validation comes first); ids of items that resolve are spelled out
                // only where the table does not know them (third stage below: every name IN a table passed the pattern when it was interned)
                if (err != ACL_ERR_INVALID_ARGUMENT && (p.rid.empty() || p.sid.empty() || (!p.ok && (!valid_object_id(p.rid) || !valid_object_id(p.sid))))) {
                    p.ok = false;
                    err = ACL_ERR_INVALID_ARGUMENT;
                }
                if (!p.ok) {
                    out[i] = acl_item_t{kDeadType, 0, 0, kDeadType, 0, 0};
                    mybad.emplace_back((uint32_t)i, err);
                    continue;
                }
                p.rt = m.rti; p.st = m.sti; p.pm = m.pmi; p.sr = m.sri;
                const Pending *prev = i > g0 && pend[i - g0 - 1].ok ? &pend[i - g0 - 1] : (i == g0 && have_last ? &last : nullptr);
                static const bool kRepeat = !getenv("ACL_INTERN_REPEAT") || atoi(getenv("ACL_INTERN_REPEAT")) != 0;  // (A/B knob)
                p.same_res = kRepeat && prev && prev->rt == p.rt && same_name(prev->rid, p.rid);
                p.same_sub = kRepeat && prev && prev->st == p.st && same_name(prev->sid, p.sid);
                p.hit_res = !p.same_res && cached(i, F_RID, p.rt, &p.kr, &p.res);  // (packed requests: this dictionary entry was resolved earlier in the call)
                p.hit_sub = !p.same_sub && cached(i, F_SID, p.st, &p.ks, &p.sub);
                if (p.hit_res) p.same_res = false;
                if (p.hit_sub) p.same_sub = false;
                if (p.hit_res && p.hit_sub) continue;
                if (!p.same_res && !p.hit_res) {
                    p.hr = ObjectTable::hash_of(p.rid);
                    h->store.objects(p.rt).prefetch(p.hr);
                }
                if (!p.same_sub && !p.hit_sub) {
                    p.hs = ObjectTable::hash_of(p.sid);
                    h->store.objects(p.st).prefetch(p.hs);
                }
            }
            for (size_t i = g0; i < g1; i++) {
                const Pending &p = pend[i - g0];
                if (!p.ok) continue;
                if (!p.same_res && !p.hit_res) h->store.objects(p.rt).prefetch_name(p.hr);
                if (!p.same_sub && !p.hit_sub) h->store.objects(p.st).prefetch_name(p.hs);
            }
            for (size_t i = g0; i < g1; i++) {
                Pending &p = pend[i - g0];
                if (!p.ok) continue;
                // unknown object ids have no relationships: sentinels above every dense id, equal only when
                // resource and subject are the same (unknown) object
                const Pending *prev = i > g0 ? &pend[i - g0 - 1] : &last;  // (same_res / same_sub were only set against an item that resolved)
                if (p.same_res) p.kr = prev->kr, p.res = prev->res;
                else if (!p.hit_res) {
                    p.kr = h->store.objects(p.rt).find_hashed(p.rid, p.hr, &p.res);
                    remember(i, F_RID, p.rt, p.kr, p.res);
                }
                if (p.same_sub) p.ks = prev->ks, p.sub = prev->sub;
                else if (!p.hit_sub) {
                    p.ks = h->store.objects(p.st).find_hashed(p.sid, p.hs, &p.sub);
                    remember(i, F_SID, p.st, p.ks, p.sub);
                }
                const bool kr = p.kr, ks = p.ks;
                uint32_t res = p.res, sub = p.sub;
                if (ids_leave_the_call) {  // (acl_resolve_bulk_v: the recycling quarantine of an unreferenced object starts over, store.hpp touch)
                    if (kr && !p.same_res) h->store.touch(p.rt, res, touch_ms);
                    if (ks && !p.same_sub) h->store.touch(p.st, sub, touch_ms);
                }
                if ((!kr && !valid_object_id(p.rid)) || (!ks && !valid_object_id(p.sid)) || p.rid == "*" || p.sid == "*") {  // (`*` never in a Check)
                    out[i] = acl_item_t{kDeadType, 0, 0, kDeadType, 0, 0};
                    mybad.emplace_back((uint32_t)i, (int32_t)ACL_ERR_INVALID_ARGUMENT);
                    continue;
                }
                if (!kr && !ks && p.rt == p.st && p.rid == p.sid) res = sub = 0xFFFFFFFEu;
                else {
                    if (!kr) res = 0xFFFFFFFDu;
                    if (!ks) sub = 0xFFFFFFFCu;
                }
                out[i] = acl_item_t{(uint16_t)p.rt, (uint16_t)p.pm, res, (uint16_t)p.st, (uint16_t)(p.sr == kNoRelation ? ACL_NO_RELATION : p.sr), sub};
            }
            have_last = pend[g1 - g0 - 1].ok;  // (the next group's first item is compared with this group's last one)
            if (have_last) last = pend[g1 - g0 - 1];
        }
        if (!mybad.empty()) {
            std::lock_guard<std::mutex> lk(bad_mu);
            bad->insert(bad->end(), mybad.begin(), mybad.end());
        }
    };
    // tens of nanoseconds per item on one thread: below 4 096 items waking the pool up (20-100 us per sleeping thread) costs more than it saves -- unless its
    // workers are still polling after the previous batch (a busy proxy's calls follow each other within that window): then from 512 items on
    bool hot = false;
    if (n >= 512 && n < 4096) {
        std::lock_guard<std::mutex> lk(h->intern_pool_mu);
        hot = h->intern_pool && h->intern_pool->awake();
    }
    if (n < 4096 && !hot) {
        run(0, n);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(h->intern_pool_mu);
        if (!h->intern_pool) h->intern_pool = new InternPool(std::min<unsigned>(std::max(2u, std::thread::hardware_concurrency()), h->intern_threads) - 1);
    }
    // threads per batch: 16 up to 32 767 items, 32 beyond (same-box A/B on a 256-thread host, profiles/r03_string_path_ab.txt: 65 536 named
    // items 135 -> 170 M decisions/s with 32; 16 384 items the same with either, 48 threads slower at both sizes)
    h->intern_pool->run(n, n >= 32768 ? 1024 : n >= 8192 ? 512 : n >= 2048 ? 128 : 64, (n >= 32768 ? h->intern_threads : std::min(16u, h->intern_threads)) - 1, run);
    if (kPieces) std::fprintf(stderr, "intern pieces of %zu items: %d threads, %.1f us inside pieces in all (longest %.1f), first began at %.1f, last ended at %.1f, back at %.1f\n", n, p_threads.load(), p_sum.load() / 1e3, p_max.load() / 1e3, p_first.load() / 1e3, p_last.load() / 1e3, (mono_ns() - tp0) / 1e3);
}

constexpr int kRouteNotTaken = -1002;
template <class Items>
static int keep_by_reverse_walk(acl_engine_t *h, const Items &its, size_t n, const uint32_t *item_off_p, size_t k_items, uint8_t *keep_out, uint8_t *pair_perm, int32_t *pair_err,
                                const CallOpts &opts, Eval *outer = nullptr);
// acl_check_bulk / acl_check_bulk_v: strings -> ids straight into the context's pinned staging, one device pass, per-item errors patched in
template <class Items>
static int check_bulk_strings(acl_engine_t *h, const Items &its, size_t n, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *o = nullptr) {
    if (!n) return h->store_only ? fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable") : ACL_OK;
    CallOpts opts;  // (cancellation / deadline: honoured while the call waits for an evaluation context, as in acl_check_bulk_ids_opts)
    if (o) {
        opts.cancel = o->cancel;
        if (o->timeout_ns > 0) opts.deadline_ns = mono_ns() + o->timeout_ns;
    }
    // A request whose pairs all name ONE plain subject, one type and one permission -- what filterItemsWithBulkPermissions sends for a list
    // (postfilter.go:67-134) -- is answered by one reverse walk + bit tests when the permission allows it (keep_by_reverse_walk, pair form); anything else
    // comes back here before a byte was written.
    {
        const int rrc = keep_by_reverse_walk(h, its, n, nullptr, n, nullptr, perm_out, err_out, opts);
        if (rrc != kRouteNotTaken) return rrc;
    }
    Eval ev;
    int rc = ev.begin(h, false, opts);
    if (rc) return rc;
    PassCtx *c = ev.c;
    std::vector<std::pair<uint32_t, int32_t>> bad;
    HIP_TRY(c->h_in.ensure(n * sizeof(acl_item_t)));
    acl_item_t *staged = (acl_item_t *)c->h_in.p;
    static const bool kTimeIt = getenv("ACL_DEBUG_STRING_TIMING") != nullptr;  // (stderr: where a string call's time goes -- tools/string_path.py)
    const int64_t t_a = kTimeIt ? mono_ns() : 0;
    {
        std::shared_lock<std::shared_mutex> nlk(h->names_mu);  // string -> id only reads the tables: concurrent callers intern in parallel
        if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
        intern_items(h, its, n, staged, &bad);
    }
    const int64_t t_b = kTimeIt ? mono_ns() : 0;
    // A request that fails the API's validation fails AS A WHOLE with InvalidArgument -- no pairs at all (validate.hpp; the reference denies
    // everything it asked on any error of the call, check.go:48-52, and fails the list response, postfilter.go:134-137).  Unknown types /
    // permissions stay per-item errors (check.go:55-60).
    for (const auto &be : bad)
        if (be.second == ACL_ERR_INVALID_ARGUMENT && !h->per_item_validation) return fail_invalid_item(h, its, be.first);
    if (bad.size() == n) {  // nothing to ask the device
        std::memset(perm_out, ACL_PERM_UNSPECIFIED, n);
    } else {
        rc = check_ids_host(h, c, staged, n, perm_out, err_out);
        if (rc) return rc;
    }
    if (kTimeIt && n >= 1024) fprintf(stderr, "[aclgpu] string call of %zu items: begin %.1f us, interning %.1f us, device pass %.1f us\n", n, 0.0, (t_b - t_a) / 1e3, (mono_ns() - t_b) / 1e3);
    for (const auto &be : bad) {
        perm_out[be.first] = ACL_PERM_UNSPECIFIED;
        err_out[be.first] = be.second;
    }
    return ACL_OK;
}

// Single-launch LookupResources over m subjects already staged in c->h_in (pinned).  Result rows go to `bitmaps` directly when the
// caller's buffer is pinned (acl_host_alloc), else through the context's pinned staging.  kTakeLevelLoop: a block outgrew its share.
static int lookup_pass_local(acl_engine *h, PassCtx *c, const DevReverse &r, uint32_t key, uint32_t target, size_t m, uint32_t *bitmaps, size_t words, size_t cw,
                             uint64_t *counts) {
    // private frontier regions: 8-byte entries carved from the context's two frontier buffers (16 B per entry there)
    uint64_t cap64 = std::min<uint64_t>(c->frontier_entries * 2 / std::max<size_t>(m, 1), 1u << 22);
    if (h->local_cap_limit) cap64 = std::min<uint64_t>(cap64, h->local_cap_limit);
    if (cap64 < 64 || words > 0xFFFFFFFFull || m > 0x7FFFFFFFull || r.nslots > kRevLdsSlots || r.nrops > kRevLdsOps) return kTakeLevelLoop;
    const bool direct = words && h->is_pinned(bitmaps, m * words * sizeof(uint32_t));
    const size_t ostride = direct ? words : cw;
    // staging: [flag (64 B)] [counts m x 8] [rows m x cw x 4]
    const size_t rows_off = 64 + m * sizeof(uint64_t);
    HIP_TRY(c->h_out.ensure(rows_off + (direct ? 0 : m * std::max<size_t>(cw, 1) * 4)));
    uint32_t *flag = (uint32_t *)c->h_out.p;
    uint64_t *h_counts = (uint64_t *)((char *)c->h_out.p + 64);
    uint32_t *h_rows = (uint32_t *)((char *)c->h_out.p + rows_off);
    *flag = 0;
    flag[15] = 0;
    void *d_sids = nullptr, *d_out = nullptr, *d_rows = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d_sids, c->h_in.p, 0));
    HIP_TRY(hipHostGetDevicePointer(&d_out, c->h_out.p, 0));
    // Result rows: written by the kernel straight into host memory (each block as it finishes), or -- rev_rows_device, A/B knob
    // ACL_REV_ROWS=device -- into a device buffer that one DMA copy brings over afterwards.
    const bool via_device = h->rev_rows_device && ostride;
    const bool spin = m <= h->spin_max && !c->timing && !via_device;
    const uint32_t done_val = spin ? next_done_val(c) : 0u;
    if (via_device) {
        HIP_TRY(c->d_rows.ensure(m * ostride));
        d_rows = c->d_rows.p;
    } else if (direct) HIP_TRY(hipHostGetDevicePointer(&d_rows, bitmaps, 0));
    else d_rows = (char *)d_out + rows_off;
    // Rows that do not fit the block's LDS (a type of more than 1 M objects; reference pkg/authz/lookups.go:49-65 asks for the whole type): the heavy terminal
    // rows are deferred to a chip-wide launch and the rows are copied / counted / cleared by a third one (kernels.hip RevDefer; ACL_REV_BIG_ROWS=0: one block
    // does it all, as in round 5 -- A/B)
    const uint32_t lds_row_words = h->rev_lds_rows ? (uint32_t)(((size_t)h->snap.slot_nobjects[target] + 31) / 32) : 0u;
    RevBigRows big;
    const size_t bm_stride = (((size_t)h->snap.slot_nobjects[target] + 127) / 128) * 128;
    // (a result slot that is a sink of the reverse graph is marked, not expanded: Snapshot::rev_sink; ACL_REV_SINK=0 at acl_open: A/B and test knob)
    const bool sink = h->rev_sink_on && h->shard.world == 1 && target < h->snap.rev_sink.size() && h->snap.rev_sink[target];
    const bool use_big = h->rev_big_rows && (lds_row_words == 0 || (size_t)lds_row_words * 4 > kRevLdsRowBytes) && cw > 0 &&
                         ((h->snap.rprogs[target].n & ~kRevRemoteBit) == 0 || sink) &&  // (a result slot nobody expands: its marks need no first-visit answer)
                         m * bm_stride <= ((size_t)2 << 30) && bm_stride <= 0xFFFFFF80ull;
    if (use_big) {
        if (c->d_big_bytes.n < m * bm_stride || !c->d_big_bytes.p) {
            HIP_TRY(c->d_big_bytes.ensure(m * bm_stride));
            c->big_bytes_zeroed = 0;
        }
        if (c->big_bytes_zeroed < m * bm_stride) {
            HIP_TRY(hipMemsetAsync(c->d_big_bytes.p, 0, m * bm_stride, c->stream));
            c->big_bytes_zeroed = m * bm_stride;
        }
        const size_t tcap = std::min<size_t>(1u << 16, std::max<size_t>(4096, ((size_t)64 << 20) / 8 / m));  // <= 64 MiB of task lists per batch
        HIP_TRY(c->d_big_tasks.ensure(m * tcap));
        HIP_TRY(c->d_big_meta.ensure(2 * m));
        if (c->d_big_counts.n < m || !c->d_big_counts.p) {
            HIP_TRY(c->d_big_counts.ensure(m));
            c->big_counts_zeroed = 0;
        }
        if (c->big_counts_zeroed < m) {
            HIP_TRY(hipMemsetAsync(c->d_big_counts.p, 0, m * sizeof(uint64_t), c->stream));
            c->big_counts_zeroed = m;
        }
        HIP_TRY(c->d_done.ensure(1));
        big = RevBigRows{c->d_big_bytes.p, (uint32_t)bm_stride, c->d_big_tasks.p, c->d_big_meta.p, c->d_big_meta.p + m, c->d_big_counts.p, (uint32_t)tcap, h->rev_defer_min};
    }
    ev_begin(c, 3);
    RevUseful useful;  // (the slots that can lead to the result slot: everything else is dead weight for this lookup)
    const bool pruned = h->rev_sink_on && h->shard.world == 1 && h->snap.rev_useful.size() >= ((size_t)target + 1) * kRevUsefulWords;
    if (pruned) std::memcpy(useful.w, h->snap.rev_useful.data() + (size_t)target * kRevUsefulWords, sizeof(useful.w));
    launch_rev_local(c->stream, r, (const uint32_t *)d_sids, (uint32_t)m, key, target | (sink ? kRevTargetSink : 0u), c->d_fbuf[0].p, c->d_fbuf[1].p, (uint32_t)cap64, (uint32_t *)d_rows, (uint32_t)ostride,
                     (uint32_t)cw, (uint64_t *)((char *)d_out + 64), (uint32_t *)d_out, lds_row_words,
                     (spin || use_big) ? c->d_done.p : nullptr, spin ? (uint32_t *)d_out + 15 : nullptr, done_val, use_big ? &big : nullptr, pruned ? &useful : nullptr);
    ev_end(c);
    if (via_device) HIP_TRY(hipMemcpyAsync(direct ? (void *)bitmaps : (void *)h_rows, c->d_rows.p, m * ostride * 4, hipMemcpyDeviceToHost, c->stream));
    // (the proxy's shape is ONE LookupResources per list request, lookups.go:65: the caller spins on the completion word -- spin_for)
    if (!(spin && spin_for(flag + 15, done_val))) HIP_TRY(hipStreamSynchronize(c->stream));
    ev_collect(c);
    if (*flag && use_big) c->big_bytes_zeroed = 0;  // (a block gave up half-way: marks of rows nobody folded may be left)
    static const bool kDebugRev = getenv("ACL_DEBUG_REV") != nullptr;  // (stderr: what the walk deferred -- tools/lookup_big_probe.py)
    if (kDebugRev && use_big) {
        std::vector<uint32_t> meta(2 * m);
        std::vector<uint64_t> tk(std::min<size_t>(big.task_cap, 4096));
        (void)hipMemcpy(meta.data(), c->d_big_meta.p, meta.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < std::min<size_t>(m, 4); i++) {
            (void)hipMemcpy(tk.data(), c->d_big_tasks.p + i * big.task_cap, std::min<size_t>(meta[i], tk.size()) * 8, hipMemcpyDeviceToHost);
            uint64_t kids = 0;
            for (size_t k = 0; k < std::min<size_t>(meta[i], tk.size()); k++) kids += tk[k] >> 32;
            fprintf(stderr, "[aclgpu] lookup %zu: %u deferred rows (%llu children in the first %zu), %u reverse levels, status %u\n", i, meta[i], (unsigned long long)kids,
                    std::min<size_t>(meta[i], tk.size()), meta[m + i], *flag);
        }
    }
    if (*flag == 2) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "a relationship row exceeds the per-task enumeration limit");
    if (*flag) {
        c->stats.overflow_retries++;
        return kTakeLevelLoop;
    }
    uint32_t levels = 0;
    for (size_t i = 0; i < m; i++) {  // count | levels walked << 56
        levels = std::max<uint32_t>(levels, (uint32_t)(h_counts[i] >> 56));
        if (counts) counts[i] = h_counts[i] & 0x00FFFFFFFFFFFFFFull;
        if (!direct) {
            uint32_t *dst = bitmaps + i * words;
            if (cw) std::memcpy(dst, h_rows + i * cw, cw * 4);
            std::fill(dst + cw, dst + words, 0u);
        }
    }
    c->stats.levels_last = levels;
    c->stats.rev_local_passes++;
    c->stats.lookup_requests += m;
    return ACL_OK;
}

// Schemas with `&` / `-`: the reverse walk only follows POSITIVE occurrences (plan_reverse.cpp), so what it marks is a superset -- the
// candidates.  The answer is the candidates the forward walk grants: one bulk Check per lookup batch, bits of everything but HAS cleared.
// (LookupResources(T, p, S) = {id : Check(T:id#p@S) = HAS}; reference call site pkg/authz/lookups.go:65.)
// A candidate whose Check ERRS (a branch beyond the dispatch depth under an `&` / `-`) fails the CALL with that item's code: the reference's
// stream ends at the first Recv error (lookups.go:75-83) and the list request with it (responsefilterer.go:196-204) -- it never sees a
// silently shorter list.  ACL_FLAG_LENIENT_LOOKUP keeps the round-4/5 behaviour (such candidates are dropped, the call succeeds).
int lookup_candidate_error(acl_engine *h, int32_t code, uint32_t id, uint32_t sid) {
    (void)h;
    return fail(code, std::string(code == ACL_ERR_DEPTH ? "LookupResources: max depth exceeded" : "LookupResources: a candidate's check failed") + " while checking candidate id " +
                          std::to_string(id) + " for subject id " + std::to_string(sid) + " (the permission holds an intersection / exclusion: candidates are confirmed by a forward Check)");
}
static int lookup_refine(acl_engine *h, PassCtx *c, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n, uint32_t *bitmaps, size_t words, size_t cw,
                         uint64_t *counts) {
    std::vector<acl_item_t> items;
    std::vector<uint8_t> answers;
    std::vector<int32_t> errs;
    const uint16_t sr = (uint16_t)(srel < 0 ? ACL_NO_RELATION : srel);
    const size_t chunk = std::max<size_t>(h->max_sub_batch, 1);
    const bool strict = !h->lenient_lookup;
    size_t i0 = 0;  // first lookup whose candidates are in `items`
    auto flush = [&](size_t i1) -> int {  // answers the candidates of lookups [i0, i1) and clears the denied ones
        if (!items.empty()) {
            answers.resize(items.size());
            if (strict) errs.assign(items.size(), 0);
            for (size_t b = 0; b < items.size(); b += chunk) {
                int rc = check_ids_host(h, c, items.data() + b, std::min(chunk, items.size() - b), answers.data() + b, strict ? errs.data() + b : nullptr);
                if (rc) return rc;
            }
            if (strict)
                for (size_t k = 0; k < items.size(); k++)
                    if (errs[k]) return lookup_candidate_error(h, errs[k], items[k].resource_id, items[k].subject_id);
            size_t k = 0;
            for (size_t i = i0; i < i1; i++) {
                uint32_t *row = bitmaps + i * words;
                for (size_t w = 0; w < cw; w++)
                    for (uint32_t m = row[w]; m; m &= m - 1, k++)
                        if (answers[k] != ACL_PERM_HAS_PERMISSION) row[w] &= ~(m & (0u - m));
            }
        }
        for (size_t i = i0; i < i1; i++)
            if (counts) counts[i] = popcount_words(bitmaps + i * words, cw);
        items.clear();
        i0 = i1;
        return ACL_OK;
    };
    for (size_t i = 0; i < n; i++) {
        const uint32_t *row = bitmaps + i * words;
        for (size_t w = 0; w < cw; w++)
            for (uint32_t m = row[w]; m; m &= m - 1)
                items.push_back(acl_item_t{(uint16_t)rtype, (uint16_t)perm, (uint32_t)(w * 32 + (size_t)__builtin_ctz(m)), (uint16_t)stype, sr, sids[i]});
        if (items.size() >= chunk)
            if (int rc = flush(i + 1)) return rc;
    }
    return flush(n);
}

// one batched reverse walk: n subjects of one class against one (type, permission); bitmaps in host memory
int lookup_batch(acl_engine *h, PassCtx *c, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n, uint32_t *bitmaps, size_t words,
                 uint64_t *counts) {
    int rc = not_sharded(h);
    if (rc) return rc;
    const Schema &sc = h->store.schema();
    const uint32_t target = (uint32_t)sc.slot(rtype, perm);
    const uint32_t key = sc.subject_key(stype, srel < 0 ? kNoRelation : srel);
    const uint32_t nobj = h->store.objects(rtype).count();
    const size_t need = (nobj + 31) / 32;
    if (words < need) return fail_detail(ACL_ERR_INVALID_ARGUMENT, kDetailBitmapTooSmall, "lookup: bitmap too small (" + std::to_string(need) + " words needed)");
    // the walk can only mark the ids the snapshot's bitmap slot covers (the build-time count plus headroom); ids interned
    // since then have no relationship in this snapshot, so their bits are zero -- never copy past the slot (advice r1)
    const size_t slot_words = ((size_t)h->snap.slot_nobjects[target] + 31) / 32;
    const size_t cw = std::min(need, slot_words);
    const size_t vwords = std::max<size_t>((size_t)((h->snap.visited_bits + 31) / 32), 1);
    const size_t group = std::max<size_t>(1, std::min<size_t>(n ? n : 1, ((size_t)1 << 28) / vwords));  // <= 1 GiB of visited bits
    for (size_t b = 0; b < n; b += group) {
        const size_t m = std::min(group, n - b);
        if (c->d_visited.n < m * vwords || !c->d_visited.p) {
            HIP_TRY(c->d_visited.ensure(m * vwords));
            c->visited_zero_words = 0;  // (fresh memory)
        }
        HIP_TRY(c->d_sids.ensure(m));
        HIP_TRY(c->h_in.ensure(m * sizeof(uint32_t)));
        std::memcpy(c->h_in.p, sids + b, m * sizeof(uint32_t));
        DevReverse r = h->dev_reverse(c, (uint32_t)vwords);
        // ONE launch for the whole group (k_rev_local: a block per lookup walks every reverse level, marks the result bits where they are
        // produced, and writes the result rows + id counts straight into host memory): no per-level launches, no status round trips, no
        // memset, no D2H copies.  A lookup that outgrows its block (private frontier region, children per level) sends the group to the
        // level loop below, which spreads it over the chip.
        if (h->rev_local) {
            // the single-launch walk takes the visited rows all zero and leaves them all zero (every block clears what it marked): one memset
            // per context and size, not one per call
            if (c->visited_zero_words < m * vwords) {
                HIP_TRY(hipMemsetAsync(c->d_visited.p, 0, m * vwords * 4, c->stream));
                c->visited_zero_words = m * vwords;
            }
            rc = lookup_pass_local(h, c, r, key, target, m, bitmaps + b * words, words, cw, counts ? counts + b : nullptr);
            if (rc == ACL_OK) continue;
            c->visited_zero_words = 0;  // a block gave up half-way (or the call failed): its marks are still there
            if (rc != kTakeLevelLoop) return rc;
        }
        c->visited_zero_words = 0;  // (the level loop below marks and does not clear)
        HIP_TRY(hipMemcpyAsync(c->d_sids.p, c->h_in.p, m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c->h_out.ensure(m * std::max<size_t>(cw, 1) * 4));
        for (int attempt = 0;; attempt++) {
            if (m > c->frontier_entries) {
                rc = alloc_frontier(h, c, m * 4);
                if (rc) return rc;
            }
            DevFrontier f = h->dev_frontier(*c);
            HIP_TRY(hipMemsetAsync(c->d_visited.p, 0, m * vwords * 4, c->stream));
            launch_rev_seed(c->stream, f, c->d_sids.p, (uint32_t)m, key);  // seeds + status block, on the device
            DevReverse rl = r;  // (the level loop walks towards the result slot too: dead ops skipped -- not on a sharded graph, whose programs are one shard's)
            if (h->rev_sink_on && h->shard.world == 1 && h->snap.rev_useful.size() >= ((size_t)target + 1) * kRevUsefulWords)
                std::memcpy(rl.useful, h->snap.rev_useful.data() + (size_t)target * kRevUsefulWords, sizeof(rl.useful));
            uint32_t levels = 0;
            hipError_t cpe = hipSuccess;
            rc = level_loop(h, c, kMaxLevels + 1, [&](uint32_t it) { launch_rev_expand(c->stream, rl, f, it); }, &levels, [&] {
                // speculative epilogue: the result rows of the target slot, one strided copy for all requests
                if (cw) {
                    hipError_t e = hipMemcpy2DAsync(c->h_out.p, cw * 4, c->d_visited.p + h->snap.slot_bit_base[target] / 32, vwords * 4, cw * 4, m,
                                                    hipMemcpyDeviceToHost, c->stream);
                    if (e != hipSuccess) cpe = e;
                }
            });
            if (rc == ACL_ERR_RESOURCE_EXHAUSTED && c->h_status[2 * kLevelSlots] == 1) {
                c->stats.overflow_retries++;
                if (c->frontier_entries >= (uint64_t)kMaxFrontierChunks * kChunk || attempt > 8) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "frontier capacity exceeded in lookup");
                int rc2 = alloc_frontier(h, c, c->frontier_entries * 4);
                if (rc2) return rc2;
                continue;
            }
            if (rc) return rc;
            if (cpe != hipSuccess) return fail(ACL_ERR_INTERNAL, std::string("lookup result copy: ") + hipGetErrorString(cpe));
            break;
        }
        c->stats.lookup_requests += m;
        for (size_t i = 0; i < m; i++) {
            uint32_t *dst = bitmaps + (b + i) * words;
            if (cw) std::memcpy(dst, (const uint32_t *)c->h_out.p + i * cw, cw * 4);
            std::fill(dst + cw, dst + words, 0u);
            if (counts) counts[b + i] = popcount_words(dst, cw);
        }
    }
    if (!h->snap.slot_nonmono.empty() && h->snap.slot_nonmono[target]) return lookup_refine(h, c, rtype, perm, stype, srel, sids, n, bitmaps, words, cw, counts);
    return ACL_OK;
}

// LookupResourcesRequest strings -> ids (lookups.go:49-62); the subject is interned so `stype:sid#srel` can be its own member
int resolve_lookup(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, int *rt_out, int *pm_out,
                   int *st_out, int *sr_out, uint32_t *sub_out) {
    if (empty(rtype) || empty(perm) || empty(stype) || empty(sid)) return fail(ACL_ERR_INVALID_ARGUMENT, "invalid LookupResourcesRequest: empty field");
    std::shared_lock<RwLock> slk(h->state_mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    const Schema &sc = h->store.schema();
    int sr = -1;
    const int rt = sc.type_of(rtype);
    {   // API validation first (validate.hpp)
        const int vs = sc.type_of(stype);
        const bool srel_given = !empty(srel) && std::strcmp(srel, "...") != 0;
        if ((rt < 0 && !valid_type_name(rtype)) || (vs < 0 && !valid_type_name(stype)) || ((rt < 0 || sc.defs[rt].find(perm) < 0) && !valid_relation_name(perm)) ||
            (srel_given && (vs < 0 || sc.defs[vs].find(srel) < 0) && !valid_relation_name(srel)) || !valid_object_id(sid))
            return fail(ACL_ERR_INVALID_ARGUMENT, "invalid LookupResourcesRequest: a field does not match the API's pattern");  // (`*` is not an object id here)
    }
    if (rt < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("object definition `") + rtype + "` not found");
    const int pm = sc.defs[rt].find(perm);
    if (pm < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("relation/permission `") + perm + "` not found under definition `" + rtype + "`");
    const int st = sc.type_of(stype);
    if (st < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("object definition `") + stype + "` not found");
    if (!empty(srel) && std::strcmp(srel, "...") != 0) {
        sr = sc.defs[st].find(srel);
        if (sr < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("relation `") + srel + "` not found under definition `" + stype + "`");
    }
    *sub_out = h->store.intern_object(st, sid);  // (a subject nobody has a relationship with: reusable after the quarantine, store.hpp)
    *rt_out = rt;
    *pm_out = pm;
    *st_out = st;
    *sr_out = sr;
    return ACL_OK;
}

static int lookup_args_ok(acl_engine *h, int rtype, int perm, int stype, int srel) {
    const Schema &sc = h->store.schema();
    if (rtype < 0 || rtype >= (int)sc.defs.size() || stype < 0 || stype >= (int)sc.defs.size() || perm < 0 ||
        perm >= (int)sc.defs[rtype].members.size() || srel >= (int)sc.defs[stype].members.size())
        return fail(ACL_ERR_FAILED_PRECONDITION, "lookup: unknown type, permission or subject relation");
    return ACL_OK;
}

int lookup_batch_call(acl_engine_t *h, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n, uint32_t *bitmaps, size_t words,
                             uint64_t *counts, const CallOpts &opts) {
    if (n && (!sids || !bitmaps)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_lookup_resources_batch: NULL buffer");
    int key_slot = -1;
    {
        std::shared_lock<RwLock> slk(h->state_mu);
        if (!h->store_only) {
            if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
            int rc = lookup_args_ok(h, rtype, perm, stype, srel);
            if (rc) return rc;
            if (srel >= 0) key_slot = h->store.schema().slot(stype, srel);
        }
    }
    Eval ev;
    int rc = ev.begin(h, true, opts, key_slot);
    if (rc) return rc;
    rc = lookup_args_ok(h, rtype, perm, stype, srel);  // (the schema may have been reloaded in between)
    if (rc) return rc;
    return lookup_batch(h, ev.c, rtype, perm, stype, srel, sids, n, bitmaps, words, counts);
}

// ---- PostFilter by ONE reverse walk (round 6).  filterItemsWithBulkPermissions (reference pkg/authz/postfilter.go:58-182) resolves every
// template of every list item for the REQUESTING USER: K x F pairs that share their subject and, per template, type and permission.  K forward walks from K
// resources ask the graph the same question LookupResources answers once: which objects of the type may this subject see.  So when all pairs of a call share
// (resource type, permission, subject type, subject id) -- a plain subject, a permission whose value no `&` / `-` can change -- the engine runs the reverse
// walk once and tests the K resource names against its row.  For a subject with FEW allowed objects it tests them the cheap way round: the row's allowed
// objects give a small set of name-hash tags that stays in cache; a pair's resource name is hashed (no memory touched but its own bytes) and looked up THERE
// -- only a tag hit goes on to the type's name table for the id and the bit, and a name the user may not see never pays the table's miss.  With MANY allowed
// every name is resolved in the table -- by the host's pass over the pairs, while the device still walks.  Every deviation -- fields that differ, a userset subject, an item the API's validation
// would refuse, a non-monotone permission, a sharded or store-only engine -- returns kRouteNotTaken BEFORE anything is written, and the caller takes the
// forward path: keep mask and error behaviour are the forward path's by construction (an unknown or unreachable resource is NO_PERMISSION there, a depth
// error is a pair error there: both drop the item, postfilter.go:162-172, as the missing bit does here).
// "No Check of (rt, pm) for a subject of type st ends at the dispatch-depth limit on this snapshot" -- what lets the PAIR form below answer a RECURSIVE permission
// (nested groups: the schema alone allows a chain of any length) from the reverse walk's row.  A Check that does not find its subject explores every path below
// its resource, whoever the subject is (the oracle's and the kernels' rule: HAS_PERMISSION wins, else a path beyond the limit is the pair's error, else NO; no
// path's length depends on who is looked for), so the set of resources with a depth error is the set a subject NOBODY IS gets one for: one forward sweep over the
// type's ids with the unknown-subject id (what intern_check_item gives a name no table holds), through the ordinary walk.  "None is deep" then holds until a
// write ADDS a path (Store::path_adds: a userset subject, an arrow's tupleset, a bulk load -- a plain grant ends the paths it is on, a removal only takes paths
// away); "some object is deep" is remembered for its own snapshot epoch only (acl_engine::snap_epoch).  The sweep costs a forward pass over the whole type
// (845 000 pods: ~1 ms), so it is only run for a graph that holds still: the first call that needs it goes forward and leaves a note, the next one that finds
// the note current sweeps.
// true: known for this snapshot -- none is deep, or *deep_out holds the deep objects' bitmap; false: not known (yet): the caller takes the forward path (a sweep that
// fails is "not known").  Caller holds state_mu shared (an Eval) and owns context c.
// *deep_out: the deep objects' bitmap when there are some (the pair form then answers their depth error per pair), empty when none is deep.
static bool no_object_is_deep(acl_engine *h, PassCtx *c, int rt, int pm, int st, size_t n_pairs, std::shared_ptr<const std::vector<uint32_t>> *deep_out) {
    static const bool kOff = getenv("ACL_DEPTH_SWEEP") && atoi(getenv("ACL_DEPTH_SWEEP")) == 0;  // (A/B and test knob)
    if (kOff) return false;
    const uint64_t epoch = h->snap_epoch, adds = h->store.path_adds();  // (writers hold state_mu exclusive: both belong to the snapshot the caller evaluates)
    const size_t count = h->store.objects(rt).count();
    if (count > std::max<size_t>((size_t)1 << 20, 64 * n_pairs)) return false;  // (a sweep of more than ~1 ms for a call that is not itself that long: forward)
    {
        std::lock_guard<std::mutex> lk(h->deep_mu);
        acl_engine::DeepKnown *k = nullptr;
        for (auto &d : h->deep_known)
            if (d.rt == rt && d.pm == pm && d.st == st) k = &d;
        if (!k) {
            if (h->deep_known.size() >= 64) h->deep_known.erase(h->deep_known.begin());
            h->deep_known.push_back(acl_engine::DeepKnown{});
            k = &h->deep_known.back();
            k->rt = rt, k->pm = pm, k->st = st;
        }
        if (k->swept && k->none && k->adds == adds) return true;  // shallow when swept, and nothing written since could have added a path
        if (k->swept && k->epoch == epoch) {                      // (this very snapshot: deep objects were found -- here is which)
            *deep_out = k->bits;
            return true;
        }
        // not known for this snapshot.  The sweep is for a graph that holds still: the first call that needs it leaves a note and goes forward, the next one
        // that finds the note still current sweeps -- current by the store's path_adds (plain grants and removals come and go without touching it), or, after a
        // sweep that FOUND deep objects, by the snapshot's epoch (only a removal can help then, and any write may be one)
        static const bool kEager = getenv("ACL_DEPTH_SWEEP") && atoi(getenv("ACL_DEPTH_SWEEP")) == 2;  // (tests: sweep at the first call)
        const bool by_epoch = k->swept && !k->none && k->adds == adds;
        const uint64_t key = by_epoch ? epoch : adds;
        if ((k->wanted != key || k->wanted_by_epoch != by_epoch) && !kEager) {
            k->wanted = key;
            k->wanted_by_epoch = by_epoch;
            return false;
        }
    }
    // (two callers may sweep the same epoch side by side: same answer, stored twice)
    const size_t chunk = std::min<size_t>(std::max<size_t>(h->max_sub_batch, 1), 262144);
    std::vector<acl_item_t> items(std::min(chunk, std::max<size_t>(count, 1)));
    std::vector<uint8_t> perm(items.size());
    std::vector<int32_t> err(items.size());
    bool none = true;
    auto bits = std::make_shared<std::vector<uint32_t>>();
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        for (size_t i = 0; i < m; i++) items[i] = acl_item_t{(uint16_t)rt, (uint16_t)pm, (uint32_t)(b + i), (uint16_t)st, (uint16_t)ACL_NO_RELATION, 0xFFFFFFFCu};
        if (check_ids_host(h, c, items.data(), m, perm.data(), err.data())) return false;
        for (size_t i = 0; i < m; i++)
            if (err[i]) {  // (any error is the depth error: an invalid item cannot happen -- the ids are the type's own)
                if (none) bits->assign((count + 31) / 32, 0u);
                none = false;
                (*bits)[(b + i) >> 5] |= 1u << ((b + i) & 31u);
            }
    }
    h->depth_sweeps.fetch_add(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(h->deep_mu);
    for (auto &d : h->deep_known)
        if (d.rt == rt && d.pm == pm && d.st == st) {
            d.swept = true;
            d.none = none;
            d.epoch = epoch;
            d.adds = adds;
            d.bits = none ? nullptr : bits;
        }
    if (!none) *deep_out = bits;
    return true;
}

template <class Items>
static int keep_by_reverse_walk(acl_engine_t *h, const Items &its, size_t n, const uint32_t *item_off_p, size_t k_items, uint8_t *keep_out, uint8_t *pair_perm, int32_t *pair_err,
                                const CallOpts &opts, Eval *outer) {
    // PAIR form (CheckBulkPermissions itself, keep_out == NULL): every pair is an "item" of its own and is answered HAS_PERMISSION / NO_PERMISSION without an
    // error -- only where no Check of the permission can end in a depth error: whatever the relationships are (Snapshot::slot_deep says so of the schema), or, for a
    // recursive permission, on THIS snapshot (no_object_is_deep) -- because the row's missing bit cannot tell "no" from "gave up at the depth limit", which the
    // forward path reports per pair.
    struct PairRanges {
        const uint32_t *off;
        size_t operator[](size_t i) const { return off ? off[i] : i; }
    } const item_off{item_off_p};
    const bool pair_form = keep_out == nullptr;
    auto emit = [&](size_t it, uint8_t all) {
        if (!pair_form) keep_out[it] = all;
        else {
            pair_perm[it] = all ? ACL_PERM_HAS_PERMISSION : ACL_PERM_NO_PERMISSION;
            pair_err[it] = 0;
        }
    };
    static const size_t kMin = [] {
        const char *e = getenv("ACL_KEEP_ROUTE_MIN");  // (A/B and test knob; 0 switches the route off)
        return e ? (size_t)std::max(0, atoi(e)) : (size_t)512;
    }();
    if (!kMin || n < kMin || h->store_only || h->shard.world > 1) return kRouteNotTaken;
    if (!k_items || item_off[0] != 0 || item_off[k_items] != n) return kRouteNotTaken;  // (pairs outside every item: the forward path checks them all the same)
    {   // a look at three pairs before any lock is taken or a reverse snapshot asked for: most bulk requests that are not one user's are not at first sight
        static const int kConst[5] = {F_RT, F_PM, F_ST, F_SID, F_SR};
        for (size_t i : {(size_t)1, n / 2, n - 1})
            for (int f : kConst) {
                const char *x = its.ptr(i, f), *y = its.ptr(0, f);
                const size_t lx = its.len(i, f), ly = its.len(0, f);
                if (!(lx == ly && (x == y || (x && y && std::memcmp(x, y, lx) == 0) || (lx == 0 && (!x || !y))))) return kRouteNotTaken;
            }
    }
    int rt, pm, st;
    uint32_t sub = 0;
    bool sub_known = false;
    // (state_mu -- the evaluation's -- then names_mu: engine_internal.hpp's order.  The evaluation begins BEFORE the call's constants are resolved: no schema
    //  reload can come between the ids taken here and the walk that uses them.  The names stay locked while the device walks: the pass resolves names then.)
    // (outer: the caller's evaluation -- a K x F call answers its F templates one after the other on ONE snapshot, keep_by_reverse_walks)
    Eval own;
    Eval &ev = outer ? *outer : own;
    if (!outer) {
        int rc = ev.begin(h, true, opts);
        if (rc) return rc;
    }
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    {   // the call's constants, from item 0
        if (!h->store.has_schema()) return kRouteNotTaken;
        NameMemo m;
        int32_t err = 0;
        if (!intern_names(h->store.schema(), its, 0, m, &err) || m.sri != kNoRelation) return kRouteNotTaken;
        rt = m.rti, pm = m.pmi, st = m.sti;
        const char *u = its.ptr(0, F_SID);
        const std::string_view sid = u ? std::string_view(u, its.len(0, F_SID)) : std::string_view();
        if (sid.empty() || sid == "*" || !valid_object_id(sid)) return kRouteNotTaken;
        sub_known = h->store.objects(st).find(sid, &sub);
        if (sub_known) h->store.touch(st, sub);  // (the id leaves the names lock: its recycling quarantine starts over, store.hpp)
    }
    auto pool = [&](bool create = true) -> InternPool * {
        std::lock_guard<std::mutex> lk(h->intern_pool_mu);
        if (!h->intern_pool && create) h->intern_pool = new InternPool(std::min<unsigned>(std::max(2u, std::thread::hardware_concurrency()), h->intern_threads) - 1);
        return h->intern_pool;
    };
    const unsigned workers = (n >= 32768 ? h->intern_threads : std::min(16u, h->intern_threads)) - 1;
    // ---- the reverse walk (none for a subject no table knows: it has no relationships, and without a subject relation it is nobody's member) AND, while the
    // device walks, the host's pass over the pairs: constants compared with item 0 (by pointer, then by content), the resource name validated and hashed.
    // The walk is a launch, 30-90 us of kernel and the row's way back; the pass is ~17 ns per pair and thread (30 when it resolves the names too): they
    // overlap, the caller waiting for the device, the pool's workers on the pairs (the caller joins them when the row is back).
    static const bool kTrace = getenv("ACL_DEBUG_KEEP") != nullptr;  // (phase times of the route on stderr)
    const auto t_0 = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count(); };
    double us_walk = 0, us_a = 0, us_fill = 0;
    std::atomic<uint64_t> tr_sum{0}, tr_first{~0ull}, tr_last{0}, tr_test{0};  // (pass chunks: ns inside them, when the first began, when the last ended)
    static thread_local std::vector<uint64_t> hv_buf;  // (per calling thread: 512 KB of fresh pages per 65 536-item call cost more than the pass itself)
    if (hv_buf.size() < n) hv_buf.resize(n);
    uint64_t *hv = hv_buf.data();
    // ... and, while the device still walks (or once it is known that MANY objects are allowed: every name goes to the table then), the pass also RESOLVES the
    // names it has just hashed -- one block of 64 behind the block whose slots it asks for, the names' bytes still in this thread's cache -- so that the test
    // after the walk is a bit test per pair.  A walk that is back with FEW allowed objects stops that: the rest is tested through the tags, no table involved.
    constexpr uint32_t kUnresolved = 0xFFFFFFFEu, kAbsent = 0xFFFFFFFFu;
    static thread_local std::vector<uint32_t> idv_buf;
    if (idv_buf.size() < n) idv_buf.resize(n);
    uint32_t *idv = idv_buf.data();
    // FEW allowed objects: so few that hashing THEIR names (a dependent miss or three each: id -> name -> bytes) is cheaper than sending the call's names to the
    // table -- a thirty-second of the pairs (a power user with 10 000 allowed pods among 65 536 pairs took 0.32 ms through the tags, 0.13 through the table)
    // (pair form on a snapshot with DEEP objects -- a cycle of groups behind some resources: no_object_is_deep hands over their bitmap; a pair whose bit is missing
    //  answers the depth error when its resource is one of them, so every name goes to the table: never "few")
    std::shared_ptr<const std::vector<uint32_t>> deep;
    const uint32_t *dbits = nullptr;
    size_t dwords = 0;
    auto is_few = [n, &dbits](uint64_t allowed) { return !dbits && allowed != 0 && allowed <= std::max<uint64_t>(64, n / 32); };
    std::atomic<bool> walk_done{false}, many_a{false}, resolved_any{false};
    // (a user's reach does not change from one list request to the next: a subject last seen with FEW allowed objects is not resolved for while the device
    //  walks -- a third of the pass's work, wasted, for the proxy's ordinary user; one never seen, or seen with MANY, is)
    // (the hint: 58 bits of the key's hash | 1 + the bit width of the allowed count the last walk returned, 0 = never seen)
    const uint64_t seen_key = (((uint64_t)sub * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)rt << 40) ^ ((uint64_t)pm << 24) ^ ((uint64_t)st << 8)) | 63ull;
    std::atomic<uint64_t> &seen_slot = h->keep_seen[(seen_key >> 20) & 255u];
    const uint64_t seen_was = seen_slot.load(std::memory_order_relaxed);
    const bool seen_known = (seen_was | 63ull) == seen_key && (seen_was & 63ull) != 0;
    const uint64_t seen_count = seen_known ? ((1ull << ((seen_was & 63ull) - 1)) >> 1) : 0;  // (a lower bound of what the last walk allowed)
    const bool guess_few = seen_known && (seen_count == 0 || is_few(seen_count));
    // A SHORT list for a subject who reaches thousands of objects: the walk (60-90 us: it marks all of them) costs more than the forward path's pass over
    // the few pairs (1 024 pairs: 0.10 against 0.06 ms).  Not taken then -- except every sixteenth time, so that the hint follows a user whose reach shrinks.
    if (sub_known && seen_count >= 4096 && n < 8192 && (h->keep_route_skips.fetch_add(1, std::memory_order_relaxed) & 15u) != 15u) return kRouteNotTaken;
    static const bool kResolveInPass = !getenv("ACL_KEEP_RESOLVE") || atoi(getenv("ACL_KEEP_RESOLVE")) != 0;  // (A/B knob)
    std::vector<uint32_t> row;
    uint64_t count = 0;
    std::atomic<int> outcome{0};  // 0 fine; 1: not a uniform call after all / an item the forward path must judge -> not taken
    int walk_rc = ACL_OK;
    const ObjectTable &tab = h->store.objects(rt);
    {
        const uint32_t target = (uint32_t)h->store.schema().slot(rt, pm);
        // (the pair form also for a subject no table knows: a walk through a cycle of groups ends at the depth limit whoever is looked for)
        if (pair_form && (h->snap.slot_deep.size() <= target || (h->snap.slot_deep[target] && !no_object_is_deep(h, ev.c, rt, pm, st, n, &deep)))) return kRouteNotTaken;
        dbits = deep ? deep->data() : nullptr;
        dwords = deep ? deep->size() : 0;
        if (sub_known) {
            if (!h->snap.slot_nonmono.empty() && h->snap.slot_nonmono[target]) return kRouteNotTaken;
            const size_t words = ((size_t)h->store.objects(rt).count() + 31) / 32;
            row.assign(std::max<size_t>(words, 1), 0u);
        }
        const std::function<void()> walk = [&] {
            const auto t_w = std::chrono::steady_clock::now();
            if (sub_known) walk_rc = lookup_batch(h, ev.c, rt, pm, st, -1, &sub, 1, row.data(), row.size(), &count);
            us_walk = us_since(t_w);
            many_a.store(dbits || (count != 0 && !is_few(count)), std::memory_order_relaxed);
            walk_done.store(true, std::memory_order_release);
            if (sub_known && !walk_rc) {
                const unsigned width = count ? 64u - (unsigned)__builtin_clzll(count) : 0u;
                seen_slot.store((seen_key & ~63ull) | std::min(62u, 1u + width), std::memory_order_relaxed);
            }
        };
        // (both passes go over ITEMS, each worker through its items' pairs [item_off[a], item_off[b]): the keep bytes are then written where they are computed)
        const std::function<void(size_t, size_t)> pass = [&](size_t a, size_t b) {
            static const int kConst[5] = {F_RT, F_PM, F_ST, F_SID, F_SR};
            struct ChunkTrace {
                std::chrono::steady_clock::time_point t0, c0;
                std::atomic<uint64_t> *sum, *first, *last;
                ~ChunkTrace() {
                    if (!sum) return;
                    const auto now = std::chrono::steady_clock::now();
                    sum->fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - c0).count());
                    const uint64_t st = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(c0 - t0).count(), en = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - t0).count();
                    for (uint64_t v = first->load(); st < v && !first->compare_exchange_weak(v, st);) {}
                    for (uint64_t v = last->load(); en > v && !last->compare_exchange_weak(v, en);) {}
                }
            } ct{t_0, std::chrono::steady_clock::now(), kTrace ? &tr_sum : nullptr, &tr_first, &tr_last};
            // Blocks of 64 items, stage by stage: the offsets; every pair's constant fields and resource id judged; the ids hashed into a LOCAL array; the hashes
            // copied out and their table slots asked for.  (Fused into one loop per pair -- judge, hash, store -- the same work measured 85 cycles per pair
            // instead of 50 on the box's EPYC, 30 of them on the store into the shared array between a pair's loads: rdtsc per stage, round 6.)
            constexpr size_t kBlock = 64;
            uint64_t loc[kBlock];
            uint32_t rid_loc[kBlock];
            size_t prev_lo = 0, prev_le = 0;  // the block whose slots were asked for last: resolved while the next block's are on their way
            auto resolve_prev = [&] {
                if (prev_le == prev_lo) return;
                const bool back = walk_done.load(std::memory_order_acquire);
                if (kResolveInPass && (back ? many_a.load(std::memory_order_relaxed) : !guess_few)) {
                    if (!resolved_any.load(std::memory_order_relaxed)) resolved_any.store(true, std::memory_order_relaxed);
                    for (size_t i = prev_lo; i < prev_le; i++) {
                        uint32_t id;
                        rid_loc[i - prev_lo] = tab.find_hashed(std::string_view(its.ptr(i, F_RID), its.len(i, F_RID)), hv[i], &id) ? id : kAbsent;
                    }
                    for (size_t i = prev_lo; i < prev_le; i++) idv[i] = rid_loc[i - prev_lo];
                }
                prev_lo = prev_le = 0;
            };
            for (size_t g = a; g < b; g += kBlock) {
                if (outcome.load(std::memory_order_relaxed)) return;
                const size_t ge = std::min(b, g + kBlock);
                bool fine = true;
                for (size_t it = g; it < ge; it++) fine &= item_off[it] <= item_off[it + 1] && item_off[it + 1] <= n;  // (else not an ascending offset array: the caller's check says so)
                if (!fine) {
                    outcome.store(1, std::memory_order_relaxed);
                    return;
                }
                for (size_t lo = item_off[g], hi = item_off[ge]; lo < hi; lo += kBlock) {
                    const size_t le = std::min(hi, lo + kBlock);
                    for (size_t i = lo; i < le; i++) {
                        // (the usual case first: the shim points every pair's constant fields at the same strings / dictionary entries -- no bytes compared)
                        bool same = true, identical = false;
                        if constexpr (std::is_same_v<Items, PackedItems>) {
                            const uint32_t *x = its.rq->items + 6 * i, *y = its.rq->items;
                            identical = ((x[F_RT] ^ y[F_RT]) | (x[F_PM] ^ y[F_PM]) | (x[F_ST] ^ y[F_ST]) | (x[F_SID] ^ y[F_SID]) | (x[F_SR] ^ y[F_SR])) == 0;
                        } else if constexpr (std::is_same_v<Items, ViewItems>) {
                            const auto *x = &its.it[i].resource_type, *y = &its.it[0].resource_type;
                            identical = true;
                            for (int k = 0; k < 5; k++) identical &= x[kConst[k]].p == y[kConst[k]].p && x[kConst[k]].n == y[kConst[k]].n;
                        }
                        for (int k = 0; k < 5 && same && !identical; k++) {
                            const int f = kConst[k];
                            const char *x = its.ptr(i, f), *y = its.ptr(0, f);
                            const size_t lx = its.len(i, f), ly = its.len(0, f);
                            same = lx == ly && (x == y || (x && y && std::memcmp(x, y, lx) == 0) || (lx == 0 && (!x || !y)));
                        }
                        const char *r = its.ptr(i, F_RID);
                        const std::string_view rid = r ? std::string_view(r, its.len(i, F_RID)) : std::string_view();
                        fine &= same && !rid.empty() && rid != "*" && valid_object_id(rid);  // (else the forward path knows what to do with it: per-item error, whole-call failure, ...)
                    }
                    if (!fine) {
                        outcome.store(1, std::memory_order_relaxed);
                        return;
                    }
                    for (size_t i = lo; i < le; i++) loc[i - lo] = ObjectTable::hash_of(std::string_view(its.ptr(i, F_RID), its.len(i, F_RID)));
                    for (size_t i = lo; i < le; i++) {
                        hv[i] = loc[i - lo];
                        idv[i] = kUnresolved;
                        tab.prefetch2(loc[i - lo]);
                    }
                    resolve_prev();
                    prev_lo = lo;
                    prev_le = le;
                }
            }
            resolve_prev();  // (the chunk's last block: one exposed trip per 1 024 pairs)
        };
        // (the pool from 2 048 items on, from 512 when its workers are still polling after the previous call: intern_items)
        InternPool *const P = pool(k_items >= 2048);
        const bool spread = k_items >= 2048 || (k_items >= 512 && P && P->awake());
        const size_t piece = k_items >= 32768 ? 1024 : k_items >= 8192 ? 512 : k_items >= 2048 ? 128 : 64;  // (two pieces per worker: one that starts late does not make the others wait)
        if (!spread) {
            walk();
            pass(0, k_items);
        } else P->run(k_items, piece, workers, pass, &walk);
    }
    if (walk_rc) return walk_rc;
    if (outcome.load()) return kRouteNotTaken;
    us_a = us_since(t_0);
    if (!outer) ev.end();
    std::vector<uint32_t> tags;  // open addressing over the allowed objects' name tags (0 = empty; a tag of 0 is stored as 1: only costs a rare extra probe)
    uint32_t tmask = 0;
    {
        // Two ways to test a name against the row.  FEW allowed objects (is_few: at most a thirty-second of the pairs): their names' hash tags make a small set that
        // stays in cache, and only a tag hit goes on to the name table.  MANY: every name goes to the table (one miss, prefetched a group ahead) -- still
        // half of what the forward path's interning pays (it resolves the subject too) and no device pass over K items.
        const bool few = is_few(count);
        if (few) {
            size_t cap = 64;
            while (cap < 2 * count) cap <<= 1;
            tags.assign(cap, 0u);
            tmask = (uint32_t)(cap - 1);
            std::atomic<uint32_t> *at = reinterpret_cast<std::atomic<uint32_t> *>(tags.data());
            std::atomic<bool> anonymous{false};
            const std::function<void(size_t, size_t)> fill = [&](size_t wa, size_t wb) {
                for (size_t w = wa; w < wb; w++)
                    for (uint32_t mbits = row[w]; mbits; mbits &= mbits - 1) {
                        const std::string *nm = tab.name((uint32_t)(w * 32 + (size_t)__builtin_ctz(mbits)));
                        if (!nm) {
                            anonymous.store(true, std::memory_order_relaxed);
                            return;
                        }
                        uint32_t tg = (uint32_t)(ObjectTable::hash_of(*nm) >> 32);
                        tg += tg == 0u;
                        for (uint32_t q = tg & tmask;; q = (q + 1) & tmask) {
                            uint32_t seen = at[q].load(std::memory_order_relaxed);
                            if (seen == tg || (seen == 0u && at[q].compare_exchange_strong(seen, tg, std::memory_order_relaxed)) || seen == tg) break;
                        }
                    }
            };
            if (count < 256) fill(0, row.size());
            else pool()->run(row.size(), std::max<size_t>(256, row.size() / 64), workers, fill);
            if (anonymous.load()) return kRouteNotTaken;  // (anonymous ids -- bulk-loaded numeric graphs -- have no names to compare with: forward path)
        }
        us_fill = us_since(t_0);
        // ---- the pairs against the row; an item is kept when every one of its pairs is (one without pairs too: postfilter.go:145-150).  A pair the pass has
        // resolved is a bit test; one it has not goes through the tags (FEW) and, on a tag hit or with MANY allowed, to the name table -- blocks of 64 pairs, the
        // outcomes into a local array first, then into the ids' place.
        const std::function<void(size_t, size_t)> test = [&](size_t a, size_t b) {
            struct T2 {
                std::chrono::steady_clock::time_point c0;
                std::atomic<uint64_t> *sum;
                ~T2() { if (sum) sum->fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c0).count()); }
            } t2{std::chrono::steady_clock::now(), kTrace ? &tr_test : nullptr};
            if (few && !resolved_any.load(std::memory_order_relaxed)) {  // (nothing was resolved: tags only, straight into the keep bytes)
                for (size_t it = a; it < b; it++) {
                    uint8_t all = 1;
                    for (size_t i = item_off[it]; i < item_off[it + 1]; i++) {
                        const uint64_t hh = hv[i];
                        uint32_t tg = (uint32_t)(hh >> 32), id;
                        tg += tg == 0u;
                        bool maybe = false;
                        for (uint32_t q = tg & tmask; tags[q] != 0u && !maybe; q = (q + 1) & tmask) maybe = tags[q] == tg;
                        all &= (uint8_t)(maybe && tab.find_hashed(std::string_view(its.ptr(i, F_RID), its.len(i, F_RID)), hh, &id) && (size_t)(id >> 5) < row.size() &&
                                         ((row[id >> 5] >> (id & 31u)) & 1u));
                    }
                    emit(it, all);
                }
                return;
            }
            constexpr size_t kBlock = 64;
            const size_t p_lo = item_off[a], p_hi = item_off[b];
            uint8_t ok[kBlock];
            for (size_t lo = p_lo; lo < p_hi; lo += kBlock) {
                const size_t le = std::min(p_hi, lo + kBlock);
                if (!few)
                    for (size_t i = lo; i < le; i++)
                        if (idv[i] == kUnresolved) tab.prefetch2(hv[i]);
                for (size_t i = lo; i < le; i++) {
                    uint32_t id = idv[i];
                    if (id == kUnresolved) {
                        const uint64_t hh = hv[i];
                        bool maybe = !few;
                        if (few) {
                            uint32_t tg = (uint32_t)(hh >> 32);
                            tg += tg == 0u;
                            for (uint32_t q = tg & tmask; tags[q] != 0u && !maybe; q = (q + 1) & tmask) maybe = tags[q] == tg;
                        }
                        // (the name table has the last word: id, then the row's bit)
                        if (!maybe || !tab.find_hashed(std::string_view(its.ptr(i, F_RID), its.len(i, F_RID)), hh, &id)) id = kAbsent;
                    }
                    ok[i - lo] = id != kAbsent && (size_t)(id >> 5) < row.size() && ((row[id >> 5] >> (id & 31u)) & 1u);
                    if (dbits && !ok[i - lo] && id != kAbsent && (size_t)(id >> 5) < dwords && ((dbits[id >> 5] >> (id & 31u)) & 1u)) ok[i - lo] = 2;  // (gave up at the depth limit)
                }
                for (size_t i = lo; i < le; i++) idv[i] = ok[i - lo];
            }
            for (size_t it = a; it < b; it++) {
                if (pair_form && idv[it] == 2u) {  // (pair form: item it IS pair it)
                    pair_perm[it] = 0;
                    pair_err[it] = ACL_ERR_DEPTH;
                    continue;
                }
                uint8_t all = 1;
                for (size_t i = item_off[it]; i < item_off[it + 1]; i++) all &= (uint8_t)idv[i];
                emit(it, all);
            }
        };
        if (!count && !dbits) {
            for (size_t it = 0; it < k_items; it++) emit(it, item_off[it] == item_off[it + 1]);
        } else if (k_items < 512 || !pool(k_items >= 2048)) test(0, k_items);
        else pool()->run(k_items, k_items >= 32768 ? 1024 : k_items >= 8192 ? 512 : k_items >= 2048 ? 128 : 64, workers, test);  // (its workers polled through the walk)
    }
    if (!outer) h->keep_route_calls.fetch_add(1, std::memory_order_relaxed);  // (a K x F call counts its F walks once all of them have answered)
    if (kTrace) std::fprintf(stderr, "keep route: n %zu allowed %llu | walk %.1f us | walk + pass done at %.1f (chunks: %.1f us in all, first began at %.1f, last ended at %.1f) | tags at %.1f | end %.1f (test chunks: %.1f us in all)\n", n, (unsigned long long)count, us_walk, us_a, tr_sum.load() / 1e3, tr_first.load() / 1e3, tr_last.load() / 1e3, us_fill, us_since(t_0), tr_test.load() / 1e3);
    return ACL_OK;
}

// K items x F templates (postfilter.go:86-119: every PostFilter of every matching rule, resolved per item -- F = 2 ... 4 here): pair j of every item comes from
// template j, so the pairs at positions j, j + F, j + 2 F, ... are a one-template call of their own.  Each is answered by its reverse walk (EveryNth: a view of
// the call's items) under ONE evaluation -- one snapshot for the whole request, as the forward path has -- and an item is kept when every template keeps it.
// Anything else -- items with different numbers of pairs, a template position whose pairs do not share type / permission / subject, a route that declines --
// returns kRouteNotTaken before keep_out is touched.
template <class Items>
struct EveryNth {
    const Items &base;
    size_t stride, phase;
    static constexpr bool kHasLen = Items::kHasLen;
    const char *ptr(size_t i, int f) const { return base.ptr(i * stride + phase, f); }
    size_t len(size_t i, int f) const { return base.len(i * stride + phase, f); }
};
template <class Items>
static int keep_by_reverse_walks(acl_engine_t *h, const Items &its, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    if (!k_items || n % k_items || h->store_only || h->shard.world > 1) return kRouteNotTaken;
    const size_t F = n / k_items;
    if (F < 2 || F > 4 || k_items < 512) return kRouteNotTaken;
    for (size_t i = 0; i <= k_items; i++)
        if (item_off[i] != i * F) return kRouteNotTaken;
    Eval ev;
    int rc = ev.begin(h, true, CallOpts());
    if (rc) return rc;
    {   // F walks must beat ONE forward pass over K F pairs: over a type of a few hundred thousand objects a walk is 20-40 us and they do from a few hundred
        // items on; over C4's 845 000 pods a walk is 30-85 us and two of them (0.21 ms at 16 384 items) beat the forward string path (0.27 ms) from there on
        // (profiles/r06_keep_route.txt) -- such types from 16 384 items on
        std::shared_lock<std::shared_mutex> nlk(h->names_mu);
        if (!h->store.has_schema()) return kRouteNotTaken;
        size_t biggest = 0;
        for (size_t j = 0; j < F; j++) {
            const char *t = its.ptr(j, F_RT);
            const int rt = t ? h->store.schema().type_of(std::string(t, its.len(j, F_RT))) : -1;
            if (rt < 0) return kRouteNotTaken;
            biggest = std::max<size_t>(biggest, h->store.objects(rt).count());
        }
        static const size_t kBigMin = getenv("ACL_KEEP_MULTI_MIN") ? (size_t)std::max(0, atoi(getenv("ACL_KEEP_MULTI_MIN"))) : (size_t)16384;  // (A/B knob)
        if (biggest > 262144 && k_items < kBigMin) return kRouteNotTaken;
    }
    std::vector<uint8_t> kj(F * k_items);
    for (size_t j = 0; j < F; j++) {
        const EveryNth<Items> view{its, F, j};
        rc = keep_by_reverse_walk(h, view, k_items, nullptr, k_items, kj.data() + j * k_items, nullptr, nullptr, CallOpts(), &ev);
        if (rc) return rc;  // (kRouteNotTaken among them)
    }
    ev.end();
    h->keep_route_calls.fetch_add(F, std::memory_order_relaxed);
    for (size_t i = 0; i < k_items; i++) {
        uint8_t all = 1;
        for (size_t j = 0; j < F; j++) all &= kj[j * k_items + i];
        keep_out[i] = all;
    }
    return ACL_OK;
}

template <class Items>
static int check_bulk_keep_strings(acl_engine_t *h, const Items &its, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out, const char *who) {
    int rc = keep_by_reverse_walk(h, its, n, item_off, k_items, keep_out, nullptr, nullptr, CallOpts());  // (checks the offsets it uses as it goes, in parallel; anything irregular comes back here)
    if (rc == kRouteNotTaken) rc = keep_by_reverse_walks(h, its, n, item_off, k_items, keep_out);  // (K x F: one walk per template)
    if (rc != kRouteNotTaken) return rc;
    for (size_t i = 0; i < k_items; i++)
        if (item_off[i] > item_off[i + 1] || item_off[i + 1] > n) return fail(ACL_ERR_INVALID_ARGUMENT, std::string(who) + ": item_off must ascend and end within n");
    std::vector<uint8_t> perm(std::max<size_t>(n, 1));
    std::vector<int32_t> err(std::max<size_t>(n, 1));
    rc = check_bulk_strings(h, its, n, perm.data(), err.data());
    if (rc) return rc;
    for (size_t i = 0; i < k_items; i++) {
        bool all = true;  // pair error or anything but HAS_PERMISSION drops the item: postfilter.go:162-172
        for (uint32_t j = item_off[i]; j < item_off[i + 1]; j++) all = all && !err[j] && perm[j] == ACL_PERM_HAS_PERMISSION;
        keep_out[i] = all ? 1 : 0;
    }
    return ACL_OK;
}
static int packed_ok(acl_engine_t *h, const acl_packed_request_t *rq, const char *who) {
    if (!rq || (rq->n_items && (!rq->items || !rq->offsets || !rq->bytes))) return fail(ACL_ERR_INVALID_ARGUMENT, std::string(who) + ": NULL buffer");
    const size_t n6 = rq->n_items * 6;
    // (one pass over 24 bytes per item: nothing below reads a string through an index it has not seen; large requests share it out)
    std::atomic<uint32_t> worst_a{0};
    const std::function<void(size_t, size_t)> scan = [&](size_t a, size_t b) {
        uint32_t wv = 0;
        for (size_t k = a; k < b; k++) {
            const uint32_t v = rq->items[k];
            wv = std::max(wv, v == ACL_PACKED_NONE ? 0u : v + 1u);
        }
        uint32_t seen = worst_a.load(std::memory_order_relaxed);
        while (wv > seen && !worst_a.compare_exchange_weak(seen, wv, std::memory_order_relaxed)) {
        }
    };
    if (rq->n_items < 16384) {
        scan(0, n6);
    } else {
        {
            std::lock_guard<std::mutex> lk(h->intern_pool_mu);
            if (!h->intern_pool) h->intern_pool = new InternPool(std::min<unsigned>(std::max(2u, std::thread::hardware_concurrency()), h->intern_threads) - 1);
        }
        h->intern_pool->run(n6, 6 * 4096, std::min(16u, h->intern_threads) - 1, scan);
    }
    const uint32_t worst = worst_a.load();
    if (worst > rq->n_strings) return fail(ACL_ERR_INVALID_ARGUMENT, std::string(who) + ": a dictionary index lies beyond n_strings");
    return ACL_OK;  // (an absent member -- ACL_PACKED_NONE -- is the pair's own InvalidArgument, as a {NULL, 0} view is)
}
int check_bulk_packed_call(acl_engine_t *h, const acl_packed_request_t *rq, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *opts) {
    if (int rc = packed_ok(h, rq, "acl_check_bulk_packed")) return rc;
    if (rq->n_items && (!perm_out || !err_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_packed: NULL buffer");
    return check_bulk_strings(h, PackedItems{rq}, rq->n_items, perm_out, err_out, opts);
}
int check_bulk_keep_packed_call(acl_engine_t *h, const acl_packed_request_t *rq, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    if (int rc = packed_ok(h, rq, "acl_check_bulk_keep_packed")) return rc;
    if (k_items && (!item_off || !keep_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_packed: NULL buffer");
    return check_bulk_keep_strings(h, PackedItems{rq}, rq->n_items, item_off, k_items, keep_out, "acl_check_bulk_keep_packed");
}
int check_bulk_keep_cstr_call(acl_engine_t *h, const acl_check_item_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    if ((n && !items) || (k_items && (!item_off || !keep_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep: NULL buffer");
    return check_bulk_keep_strings(h, CStrItems{items}, n, item_off, k_items, keep_out, "acl_check_bulk_keep");
}
int check_bulk_keep_v_call(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    if ((n && !items) || (k_items && (!item_off || !keep_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_v: NULL buffer");
    return check_bulk_keep_strings(h, ViewItems{items}, n, item_off, k_items, keep_out, "acl_check_bulk_keep_v");
}

int lookup_opts_call(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap_out,
                     size_t bitmap_words, uint64_t *count_out, const CallOpts &opts) {
    int rt, pm, st, sr;
    uint32_t sub;
    int rc = resolve_lookup(h, rtype, perm, stype, sid, srel, &rt, &pm, &st, &sr, &sub);
    if (rc) return rc;
    return lookup_batch_call(h, rt, pm, st, sr, &sub, 1, bitmap_out, bitmap_words, count_out, opts);
}

}  // namespace aclint

bool acl_engine::is_pinned(const void *p, size_t bytes) {
    if (!p) return false;
    std::lock_guard<std::mutex> lk(pinned_mu);
    const uintptr_t a = (uintptr_t)p;
    for (const auto &r : pinned)
        if (a >= r.first && a + bytes <= r.first + r.second) return true;
    return false;
}

extern "C" {

const char *acl_last_error(void) { return g_last_error.c_str(); }

int acl_open(const acl_config_t *cfg, acl_engine_t **out) { return acl_open_replicas(cfg, nullptr, 0, out); }

int acl_open_replicas(const acl_config_t *cfg, const int32_t *devices, uint32_t n_devices, acl_engine_t **out) {
    if (!out) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_open: out is NULL");
    *out = nullptr;
    if (cfg && (cfg->flags & ACL_FLAG_STORE_ONLY)) {
        auto *so = new acl_engine();
        so->store_only = true;
        so->per_item_validation = (cfg->flags & ACL_FLAG_PER_ITEM_VALIDATION) != 0;
        so->lenient_lookup = (cfg->flags & ACL_FLAG_LENIENT_LOOKUP) != 0;
        if (const char *ev = getenv("ACL_RAW_INTERN")) so->raw_intern = atoi(ev) != 0;  // (test knob, see below)
        if (const char *ev = getenv("ACL_INTERN_THREADS")) so->intern_threads = (unsigned)std::min(64, std::max(2, atoi(ev)));  // (A/B knob, see below)
        batcher_create(so);
        *out = so;
        return ACL_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(ACL_ERR_UNAVAILABLE, "acl_open: no HIP device available (this engine has no CPU evaluation path)");
    auto h = std::make_unique<acl_engine>();
    h->per_item_validation = cfg && (cfg->flags & ACL_FLAG_PER_ITEM_VALIDATION) != 0;
    h->lenient_lookup = cfg && (cfg->flags & ACL_FLAG_LENIENT_LOOKUP) != 0;
    // the replicas: the device list of acl_open_replicas, else ACL_DEVICES="0,1,2,3" (a device may be named more than once: N logical replicas
    // on one GPU -- what the one-GPU test boxes exercise), else the one device of the config
    std::vector<int> list;
    if (devices && n_devices) list.assign(devices, devices + n_devices);
    else if (const char *ev = getenv("ACL_DEVICES")) {
        for (const char *q = ev; *q;) {
            char *end = nullptr;
            const long v = std::strtol(q, &end, 10);
            if (end == q) break;
            list.push_back((int)v);
            q = *end == ',' ? end + 1 : end;
        }
    }
    if (list.empty()) list.push_back(cfg ? cfg->device : -1);
    if (list.size() > 64) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_open: at most 64 replicas");
    for (size_t i = 0; i < list.size(); i++) {
        int dev = list[i];
        if (dev < 0) {
            if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        }
        if (dev >= ndev) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_open: device ordinal out of range");
        auto d = std::make_unique<DevState>();
        d->device = dev;
        d->index = (int)i;
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->up_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return fail(ACL_ERR_UNAVAILABLE, std::string("acl_open: ") + hipGetErrorString(e));
        d->grid_blocks = expand_grid_blocks(dev);
        d->local_blocks = local_grid_blocks(dev, 2048);  // (refined per snapshot: ensure_snapshot)
        d->local_blocks_wide = local_grid_blocks(dev, 2048, true);
        h->devs.push_back(std::move(d));
    }
    if (const char *ev = getenv("ACL_SPIN_MAX")) h->spin_max = (uint32_t)std::max(0, atoi(ev));  // A/B knob
    if (const char *ev = getenv("ACL_LOCAL_WIDE_MIN")) h->local_wide_min = (uint32_t)std::max(0, atoi(ev));  // A/B knob
    if (const char *ev = getenv("ACL_REV_LOCAL")) h->rev_local = atoi(ev) != 0;
    if (const char *ev = getenv("ACL_REV_ROWS")) h->rev_rows_device = !std::strcmp(ev, "device");
    if (const char *ev = getenv("ACL_REV_LDS_ROWS")) h->rev_lds_rows = atoi(ev) != 0;
    if (const char *ev = getenv("ACL_REV_SINK")) h->rev_sink_on = atoi(ev) != 0;
    if (const char *ev = getenv("ACL_REV_DEFER_MIN")) h->rev_defer_min = (uint32_t)std::max(1, atoi(ev));  // test knob: small graphs defer too
    if (const char *ev = getenv("ACL_REV_BIG_ROWS")) h->rev_big_rows = atoi(ev) != 0;  // A/B knob: 0 = rows beyond the LDS are walked, copied and cleared by ONE block (round 5)
    if (const char *ev = getenv("ACL_SHARD_A2A")) h->shard_a2a = atoi(ev) != 0;
    if (const char *ev = getenv("ACL_COMPACTION_SLACK")) h->compaction_slack = (uint64_t)std::max(0, atoi(ev));  // test knob (tools/fuzz_gpu.py --compact-early): small graphs compact too
    if (const char *ev = getenv("ACL_HOSTMAP_MAX")) h->hostmap_max = (uint32_t)std::max(0, atoi(ev));  // A/B knob: batches up to this size are read / answered across PCIe by the kernel itself
    if (const char *ev = getenv("ACL_INTERN_THREADS")) h->intern_threads = (unsigned)std::min(64, std::max(2, atoi(ev)));  // A/B knob: host threads of bulk string interning
    if (const char *ev = getenv("ACL_LOCAL_CAP")) h->local_cap_limit = (uint32_t)std::max(256, atoi(ev));  // test knob: forces walks to overflow
    if (const char *ev = getenv("ACL_RAW_INTERN")) h->raw_intern = atoi(ev) != 0;  // test knob: acl_intern takes any bytes (the JSON scanners' decoding tests name objects no API request could)
    if (const char *ev = getenv("ACL_HOST_SKEW_PCT")) h->host_skew_pct = (uint32_t)std::min(90, std::max(0, atoi(ev)));
    if (const char *ev = getenv("ACL_HOST_SPLIT")) h->host_split = (uint32_t)std::min(4, std::max(1, atoi(ev)));
    if (const char *ev = getenv("ACL_LOCAL_UPW")) h->local_upw = (uint32_t)std::max(1, atoi(ev));  // A/B knob: units per resident wave
    if (const char *ev = getenv("ACL_LOCAL_STATIC_PCT")) h->local_static_pct = (uint32_t)std::min(100, std::max(10, atoi(ev)));  // A/B knobs: share of a chip-filling batch
    if (const char *ev = getenv("ACL_LOCAL_DYN_UNIT")) h->local_dyn_unit = (uint32_t)std::min(256, std::max(1, atoi(ev)));       // in static units; size of the hand-out units
    if (cfg && cfg->max_sub_batch) h->max_sub_batch = cfg->max_sub_batch;
    if (cfg && cfg->frontier_entries) h->cfg_frontier_entries = cfg->frontier_entries;
    if (cfg && cfg->contexts) h->max_ctx = std::min<uint32_t>(cfg->contexts, 16);
    if (const char *ev = getenv("ACL_LOCAL_MAX")) h->local_max_items = (uint32_t)atoi(ev);  // A/B knob: 0 disables the single-launch path
    // the first context of every replica is created here, so that "out of device memory" surfaces at open
    for (auto &d : h->devs) {
        std::unique_ptr<PassCtx> c0;
        int rc = new_ctx(h.get(), d.get(), &c0, 0);
        if (rc) return rc;
        d->free_ctxs.push_back(c0.get());
        d->ctxs.push_back(std::move(c0));
        // ACL_FLAG_EAGER_CONTEXTS: every context of the pool now -- a context made on demand costs its caller (and, under pool_mu, every caller that
        // arrives meanwhile) the allocation of its frontier buffers and pinned staging, milliseconds into the first request that finds the pool busy
        for (uint32_t k = 1; cfg && (cfg->flags & ACL_FLAG_EAGER_CONTEXTS) && k < h->max_ctx; k++) {
            std::unique_ptr<PassCtx> ck;
            rc = new_ctx(h.get(), d.get(), &ck, (int)k);
            if (rc) return rc;
            d->free_ctxs.insert(d->free_ctxs.begin(), ck.get());  // (behind the first one: the pool hands out the context released last)
            d->ctxs.push_back(std::move(ck));
        }
    }
    (void)hipSetDevice(h->dev0().device);
    batcher_create(h.get());
    *out = h.release();
    return ACL_OK;
}

void acl_close(acl_engine_t *h) {
    if (!h) return;
    (void)acl_batcher_stop(h);
    async_shutdown(h);
    batcher_destroy(h);
    intern_pool_destroy(h);
    (void)acl_shard_rccl_destroy(h);
    compaction_join(h);
    if (h->store_only) {
        delete h;
        return;
    }
    {
        std::lock_guard<RwLock> lk(h->state_mu);  // waits for evaluations in flight
        for (auto &d : h->devs) {
            (void)hipSetDevice(d->device);
            d->ctxs.clear();
        }
        (void)hipSetDevice(h->dev0().device);
        h->shard_ctx.reset();
        if (h->compaction)
            for (auto &pd : h->compaction->per)
                if (pd->stream) {
                    (void)hipSetDevice(pd->device);
                    (void)hipStreamDestroy(pd->stream);
                }
        h->compaction.reset();
        std::lock_guard<std::mutex> plk(h->pinned_mu);
        for (auto &r : h->pinned) (void)hipHostFree((void *)r.first);
        h->pinned.clear();
    }
    std::vector<std::pair<int, hipStream_t>> ups;
    for (auto &d : h->devs) ups.emplace_back(d->device, d->up_stream);
    delete h;  // (frees the replicas' arrays: hipFree takes pointers of any device)
    for (auto &u : ups)
        if (u.second) {
            (void)hipSetDevice(u.first);
            (void)hipStreamDestroy(u.second);
        }
}

int acl_load_bootstrap(acl_engine_t *h, const char *schema, size_t schema_len, const char *rels, size_t rels_len) {
    std::lock_guard<RwLock> lk(h->state_mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    if (!schema) return fail(ACL_ERR_INVALID_ARGUMENT, "schema is NULL");
    compaction_join(h);  // a background build of the OLD schema's graph: wait for it and drop it (its ids mean nothing from here on)
    Status s = h->store.load_schema(std::string(schema, schema_len));
    if (!s.ok()) return fail(s);
    h->snap_valid = false;
    h->set_dev_valid(false);
    h->set_rev_uploaded(false);
    if (rels && rels_len) {
        s = h->store.load_relationship_lines(std::string(rels, rels_len));
        if (!s.ok()) return fail(s);
    }
    return ACL_OK;
}

int acl_type_id(acl_engine_t *h, const char *type) {
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    return type ? h->store.schema().type_of(type) : -1;
}
int acl_relation_id(acl_engine_t *h, int type, const char *name) {
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (!name || type < 0 || type >= (int)sc.defs.size()) return -1;
    return sc.defs[type].find(name);
}
int acl_intern(acl_engine_t *h, int type, const char *object_id, uint32_t *id_out) {
    std::shared_lock<RwLock> slk(h->state_mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (empty(object_id) || !id_out || type < 0 || type >= (int)sc.defs.size()) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_intern: bad argument");
    if (!h->raw_intern && !valid_object_id(object_id)) return fail(ACL_ERR_INVALID_ARGUMENT, std::string("acl_intern: `") + object_id + "` does not match the API's object id pattern");  // (validate.hpp: names in the tables are well-formed)
    *id_out = h->store.intern_object(type, object_id, true);  // (the caller keeps this id: never recycled)
    return ACL_OK;
}
int acl_find(acl_engine_t *h, int type, const char *object_id, uint32_t *id_out) {
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (empty(object_id) || !id_out || type < 0 || type >= (int)sc.defs.size()) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_find: bad argument");
    if (!h->store.objects(type).find(object_id, id_out)) return fail(ACL_ERR_NOT_FOUND, "object not found");
    h->store.touch(type, *id_out);  // (the id is the caller's for the length of a quarantine: an unreferenced object is not renamed under it)
    return ACL_OK;
}
int64_t acl_object_name_copy(acl_engine_t *h, int type, uint32_t id, char *buf, size_t cap) {
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return -1;
    const std::string *n = h->store.objects(type).name(id);
    if (!n) return -1;
    if (buf && cap) {
        const size_t k = std::min(n->size(), cap - 1);
        std::memcpy(buf, n->data(), k);
        buf[k] = 0;
    }
    return (int64_t)n->size();
}
// The names behind a LookupResources bitmap, a block per call (the shim's stream, lookups.go:75-83: one cgo call and one turn at the names lock per BLOCK of results,
// not per result).
int acl_bitmap_names(acl_engine_t *h, int type, const uint32_t *bitmap, size_t words, uint64_t *cursor, char *buf, size_t cap, uint32_t *ends, size_t max_names, size_t *n_out) {
    if (!cursor || !n_out || (words && !bitmap) || !buf || !ends || cap < 1024 || cap > 0xFFFFFFFFull || !max_names)
        return fail(ACL_ERR_INVALID_ARGUMENT, "acl_bitmap_names: bad argument (buf must hold at least 1024 bytes: the longest object id)");
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_bitmap_names: unknown object type");
    const ObjectTable &ot = h->store.objects(type);
    size_t n = 0, used = 0;
    uint64_t bit = *cursor;
    const uint64_t nbits = (uint64_t)words * 32;
    while (bit < nbits && n < max_names) {
        const uint32_t wv = bitmap[bit >> 5] >> (bit & 31u);
        if (!wv) {
            bit = (bit | 31u) + 1;
            continue;
        }
        bit += (uint64_t)__builtin_ctz(wv);
        const std::string *nm = ot.name((uint32_t)bit);  // (an id that lost its name meanwhile, or an anonymous bulk-loaded one: an empty name, as the per-id call's -1)
        const size_t len = nm ? nm->size() : 0;
        if (used + len > cap) break;  // (the next call starts here: cap >= the longest id, so a call always makes progress)
        if (len) std::memcpy(buf + used, nm->data(), len);
        used += len;
        ends[n++] = (uint32_t)used;
        bit++;
    }
    *cursor = bit;
    *n_out = n;
    return ACL_OK;
}
const char *acl_object_name(acl_engine_t *h, int type, uint32_t id) {
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return nullptr;
    const std::string *n = h->store.objects(type).name(id);
    return n ? n->c_str() : nullptr;
}
uint32_t acl_object_count(acl_engine_t *h, int type) {
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return 0;
    return h->store.objects(type).count();
}

int acl_write(acl_engine_t *h, const acl_update_t *ups, int n, const acl_filter_t *pre, int m, uint64_t *rev) {
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
    std::lock_guard<RwLock> lk(h->state_mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    Status s = h->store.write(u, p, rev);
    if (s.ok()) h->feed_wake();  // (acl_watch_wait)
    return s.ok() ? ACL_OK : fail(s);
}

int acl_delete_by_filter(acl_engine_t *h, const acl_filter_t *f, uint64_t *n_deleted, uint64_t *rev) {
    return acl_delete_by_filter_pre(h, f, nullptr, 0, n_deleted, rev);
}

// DeleteRelationships with OptionalPreconditions: evaluated against the pre-delete state, atomically with the delete
int acl_delete_by_filter_pre(acl_engine_t *h, const acl_filter_t *f, const acl_filter_t *pre, int n_pre, uint64_t *n_deleted, uint64_t *rev) {
    if (!f) return fail(ACL_ERR_INVALID_ARGUMENT, "filter is NULL");
    if (n_pre < 0 || (n_pre && !pre)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_delete_by_filter_pre: bad argument");
    std::vector<FilterText> p(n_pre);
    for (int i = 0; i < n_pre; i++) p[i] = to_filter(&pre[i]);
    std::lock_guard<RwLock> lk(h->state_mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    if (n_pre) {
        Status ps = h->store.check_preconditions(p);
        if (!ps.ok()) return fail(ps);
    }
    Status s = h->store.delete_by_filter(to_filter(f), n_deleted, rev);
    if (s.ok()) h->feed_wake();
    return s.ok() ? ACL_OK : fail(s);
}

int acl_read(acl_engine_t *h, const acl_filter_t *f, acl_read_cb cb, void *user) {
    if (!f || !cb) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_read: bad argument");
    std::lock_guard<RwLock> lk(h->state_mu);  // the scan settles pending bulk appends: exclusive
    Status s = h->store.read(to_filter(f), [&](const RelText &r) {
        acl_relationship_t o{r.rtype.c_str(), r.rid.c_str(), r.rel.c_str(), r.stype.c_str(), r.sid.c_str(), r.srel.c_str(), r.expires_at};
        cb(user, &o);
    });
    return s.ok() ? ACL_OK : fail(s);
}

int acl_add_edges(acl_engine_t *h, int rtype, int rel, int stype, int srel, size_t n, const uint32_t *res, const uint32_t *subj) {
    std::lock_guard<RwLock> lk(h->state_mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    Status s = h->store.add_edges(rtype, rel, stype, srel, n, res, subj);
    return s.ok() ? ACL_OK : fail(s);
}

uint64_t acl_revision(acl_engine_t *h) {
    std::shared_lock<RwLock> lk(h->state_mu);
    return h->store.revision();
}
int acl_set_now(acl_engine_t *h, int64_t t) {
    std::lock_guard<RwLock> lk(h->state_mu);
    h->store.set_now(t);
    return ACL_OK;
}
int acl_snapshot(acl_engine_t *h) {
    std::lock_guard<RwLock> lk(h->state_mu);
    return ensure_snapshot(h);  // (sets the device of every replica it uploads to)
}

int acl_replica_calls(acl_engine_t *h, uint64_t *calls_out, int32_t *devices_out, uint32_t cap) {
    std::lock_guard<std::mutex> lk(h->pool_mu);
    for (uint32_t i = 0; i < cap && i < h->devs.size(); i++) {
        if (calls_out) calls_out[i] = h->devs[i]->calls;
        if (devices_out) devices_out[i] = h->devs[i]->device;
    }
    return (int)h->devs.size();
}

void *acl_stream(acl_engine_t *h) { return (h->devs.empty() || h->dev0().ctxs.empty()) ? nullptr : (void *)h->dev0().ctxs[0]->stream; }
int acl_sync(acl_engine_t *h) {
    if (h->store_only) return fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable");
    std::vector<std::pair<int, hipStream_t>> ss;
    {
        std::lock_guard<std::mutex> lk(h->pool_mu);
        for (auto &d : h->devs)
            for (auto &c : d->ctxs) ss.emplace_back(d->device, c->stream);
    }
    for (auto &s : ss) {
        HIP_TRY(hipSetDevice(s.first));
        HIP_TRY(hipStreamSynchronize(s.second));
    }
    return ACL_OK;
}

int acl_check_bulk_ids_device(acl_engine_t *h, const void *d_items, size_t n, void *d_perm_out, void *d_err_out) {
    if (n && (!d_items || !d_perm_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_ids_device: NULL buffer");
    Eval ev;
    int rc = ev.begin(h, false, CallOpts(), -1, device_of(h, d_perm_out));  // (a replica on the device the caller's buffers live on)
    if (rc) return rc;
    rc = check_device(h, ev.c, (const uint4 *)d_items, n, (uint8_t *)d_perm_out, (int32_t *)d_err_out);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ev.c->stream));
    ev_collect(ev.c);
    return ACL_OK;
}

int acl_check_bulk_ids(acl_engine_t *h, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out) {
    return acl_check_bulk_ids_opts(h, items, n, perm_out, err_out, nullptr);
}

int acl_check_bulk_ids_opts(acl_engine_t *h, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *o) {
    if (n && (!items || !perm_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_ids: NULL buffer");
    CallOpts opts;
    if (o) {
        opts.cancel = o->cancel;
        if (o->timeout_ns > 0) opts.deadline_ns = mono_ns() + o->timeout_ns;
    }
    if (!n) return h->store_only ? fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable") : ACL_OK;
    Eval ev;
    int rc = ev.begin(h, false, opts);
    if (rc) return rc;
    return check_ids_host(h, ev.c, items, n, perm_out, err_out);
}

int acl_check_bulk(acl_engine_t *h, const acl_check_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out) {
    if (n && (!items || !perm_out || !err_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk: NULL buffer");
    return check_bulk_strings(h, CStrItems{items}, n, perm_out, err_out);
}

// the same with {pointer, length} fields: no strlen per field, and no NUL needed behind the bytes -- a cgo shim points at Go string data
int acl_check_bulk_v(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, uint8_t *perm_out, int32_t *err_out) {
    if (n && (!items || !perm_out || !err_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_v: NULL buffer");
    return check_bulk_strings(h, ViewItems{items}, n, perm_out, err_out);
}

int acl_check_bulk_packed(acl_engine_t *h, const acl_packed_request_t *req, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *opts) {
    return check_bulk_packed_call(h, req, perm_out, err_out, opts);
}
int acl_check_bulk_keep_packed(acl_engine_t *h, const acl_packed_request_t *req, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    return check_bulk_keep_packed_call(h, req, item_off, k_items, keep_out);
}
int acl_check_bulk_keep_v(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    return check_bulk_keep_v_call(h, items, n, item_off, k_items, keep_out);
}

// names -> the 16-byte items of the id entry points, in bulk and without a device pass (works on a store-only engine)
int acl_resolve_bulk_v(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, acl_item_t *out, int32_t *err_out) {
    if (n && (!items || !out || !err_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_resolve_bulk_v: NULL buffer");
    std::vector<std::pair<uint32_t, int32_t>> bad;
    {
        std::shared_lock<std::shared_mutex> nlk(h->names_mu);
        if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
        intern_items(h, ViewItems{items}, n, out, &bad, true);
    }
    if (n) std::memset(err_out, 0, n * sizeof(int32_t));
    for (const auto &be : bad) err_out[be.first] = be.second;
    return ACL_OK;
}

int acl_check_bulk_v_opts(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *opts) {
    if (n && (!items || !perm_out || !err_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_v_opts: NULL buffer");
    return check_bulk_strings(h, ViewItems{items}, n, perm_out, err_out, opts);
}

int acl_lookup_resources_batch(acl_engine_t *h, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n, uint32_t *bitmaps,
                               size_t words, uint64_t *counts) {
    return lookup_batch_call(h, rtype, perm, stype, srel, sids, n, bitmaps, words, counts, CallOpts());
}

int acl_lookup_resources_ids(acl_engine_t *h, int rtype, int perm, int stype, int srel, uint32_t sid, uint32_t *bitmap, size_t words, uint64_t *count) {
    return acl_lookup_resources_batch(h, rtype, perm, stype, srel, &sid, 1, bitmap, words, count);
}

int acl_lookup_resources(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap,
                         size_t words, uint64_t *count) {
    return lookup_opts_call(h, rtype, perm, stype, sid, srel, bitmap, words, count, CallOpts());
}

// LookupResources with an engine-owned result: the bitmap is sized by the engine at the moment of the walk, so a racing
// WriteRelationships that interns new objects of the type cannot make the caller's buffer "too small" (advice r1).
int acl_lookup_resources_alloc(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel,
                               const acl_call_opts_t *o, uint32_t **bitmap_out, size_t *words_out, uint64_t *count_out) {
    if (!bitmap_out || !words_out) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_lookup_resources_alloc: NULL output");
    *bitmap_out = nullptr;
    *words_out = 0;
    CallOpts opts;
    if (o) {
        opts.cancel = o->cancel;
        if (o->timeout_ns > 0) opts.deadline_ns = mono_ns() + o->timeout_ns;
    }
    int rt, pm, st, sr;
    uint32_t sub;
    int rc = resolve_lookup(h, rtype, perm, stype, sid, srel, &rt, &pm, &st, &sr, &sub);
    if (rc) return rc;
    for (int attempt = 0; attempt < 8; attempt++) {
        const size_t words = ((size_t)acl_object_count(h, rt) + 31) / 32 + 64;  // slack: objects interned while the walk runs
        uint32_t *bm = (uint32_t *)std::malloc(std::max<size_t>(words, 1) * sizeof(uint32_t));
        if (!bm) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "out of host memory for the result bitmap");
        rc = lookup_one_routed(h, rt, pm, st, sr, sub, bm, words, count_out, opts);
        if (rc == ACL_OK) {
            *bitmap_out = bm;
            *words_out = words;
            return ACL_OK;
        }
        std::free(bm);
        if (rc != ACL_ERR_INVALID_ARGUMENT || g_last_detail != kDetailBitmapTooSmall) return rc;  // (a typed detail, not the message's wording: ADVICE r2)
    }
    return fail(ACL_ERR_UNAVAILABLE, "lookup: the object table kept growing faster than the result bitmap");
}
void acl_free(void *p) { std::free(p); }

int acl_stats(acl_engine_t *h, acl_stats_t *out) {
    if (!out) return fail(ACL_ERR_INVALID_ARGUMENT, "out is NULL");
    uint64_t recycled;
    {
        std::shared_lock<std::shared_mutex> nlk(h->names_mu);
        recycled = h->store.ids_recycled();
    }
    std::lock_guard<std::mutex> lk(h->stats_mu);
    *out = h->stats;
    out->ids_recycled = recycled;
    out->keep_route_calls = h->keep_route_calls.load(std::memory_order_relaxed);
    out->depth_sweeps = h->depth_sweeps.load(std::memory_order_relaxed);
    return ACL_OK;
}
int acl_stats_reset(acl_engine_t *h) {
    std::lock_guard<std::mutex> lk(h->stats_mu);
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
    h->timing.store(on != 0);
    return ACL_OK;
}

}  // extern "C"
