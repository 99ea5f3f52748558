// engine_shard.cpp -- acl_shard_*: one engine = one shard of the type-hash partitioned graph.  Two ways to drive it:
//   * step protocol (acl_shard_check_begin/step/import/finish, acl_shard_lookup_*): the HOST moves the exported frontier
//     entries between shards (aclgpu/sharded.py over torch.distributed / gloo in the CPU tests);
//   * native loop (engine_shard_rccl.cpp, acl_shard_check_bulk): the whole level loop inside this library over an RCCL
//     communicator -- one fixed-capacity exchange per level, termination decided on the device.
// The protocol keeps state across calls, so it owns a context of its own (h->shard_ctx, never pooled) and serialises on shard_mu.
#include "engine_internal.hpp"

namespace aclint {

DevShard dev_shard(acl_engine *h, PassCtx *c, void *d_export, size_t cap) {
    DevShard sh;
    sh.exp = (uint4 *)d_export;
    sh.exp_count = c->d_status.p + 2 * kLevelSlots + 1;
    sh.cap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFFFu);
    sh.rank = h->shard.rank;
    sh.world = h->shard.world;
    return sh;
}

// one call of the step protocol: shard_mu + state_mu shared (+ the snapshot brought up to date when `fresh`)
int ShardCall::begin(acl_engine *h_, bool fresh, bool need_reverse, bool combine_ok) {
    h = h_;
    if (h->store_only) return fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable");
    HIP_TRY(hipSetDevice(h->dev0().device));  // (the sharded entry points run on the first replica)
    h->shard_mu.lock();
    have_mu = true;
    for (;;) {
        h->state_mu.lock_shared();
        if (!fresh || snapshot_current(h, need_reverse)) {
            locked = true;
            break;
        }
        h->state_mu.unlock_shared();
        std::lock_guard<RwLock> lk(h->state_mu);
        int rc = need_reverse ? ensure_reverse(h) : ensure_snapshot(h);
        if (rc) return rc;
    }
    // Schemas with `&` / `-` / `.all()`: Check through the native loop (acl_shard_check_bulk: cells in per-shard ranges of one global cell space,
    // round 5).  The host-driven step protocol and LookupResources on the sharded graph still refuse them.
    if (h->store.schema().has_combine && !combine_ok)
        return fail(ACL_ERR_FAILED_PRECONDITION, "a schema with intersection / exclusion is evaluated on the sharded graph by acl_shard_check_bulk only (LookupResources and the step protocol: use replicas)");
    if (!h->shard_ctx) {
        std::unique_ptr<PassCtx> nc;
        int rc = new_ctx(h, &h->dev0(), &nc, -1);
        if (rc) return rc;
        h->shard_ctx = std::move(nc);
    }
    c = h->shard_ctx.get();
    c->timing = h->timing.load(std::memory_order_relaxed);
    return ACL_OK;
}
ShardCall::~ShardCall() {
    if (c) merge_stats(h, c);
    if (locked) h->state_mu.unlock_shared();
    if (have_mu) h->shard_mu.unlock();
}

}  // namespace aclint

namespace {

// reads back the status block after a level and fills the step report
int shard_report(PassCtx *c, uint32_t iter, acl_shard_step_t *out) {
    HIP_TRY(hipMemcpyAsync(c->h_status, c->d_status.p, kStatusWords * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ev_collect(c);
    out->exported = c->h_status[2 * kLevelSlots + 1];
    out->produced = c->h_status[kLevelSlots + iter];
    out->overflow = c->h_status[2 * kLevelSlots];
    return ACL_OK;
}

int shard_ready(acl_engine *h, uint32_t iter) {
    if (iter == 0 || iter >= kLevelSlots) return fail(ACL_ERR_INVALID_ARGUMENT, "shard step: iteration out of range");
    if (!h->snap_valid || !h->dev0().dev_valid) return fail(ACL_ERR_FAILED_PRECONDITION, "shard step without acl_shard_*_begin");
    return ACL_OK;
}

int check_step(acl_engine *h, uint32_t level, void *d_has, void *d_err, void *d_export, size_t cap, bool by_dest, acl_shard_step_t *out, uint64_t *by_dest_out) {
    if (!out || !d_has || !d_err || (cap && !d_export)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_step: NULL buffer");
    ShardCall sc;
    int rc = sc.begin(h, false, false);
    if (rc) return rc;
    PassCtx *c = sc.c;
    rc = shard_ready(h, level);
    if (rc) return rc;
    if (level > kMaxLevels) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_step: level beyond the dispatch depth limit");
    if (by_dest && h->shard.world > kMaxShards) return fail(ACL_ERR_INVALID_ARGUMENT, "per-destination export supports at most 64 shards");
    HIP_TRY(hipMemsetAsync(c->d_status.p + 2 * kLevelSlots + 1, 0, (1 + kMaxShards) * sizeof(uint32_t), c->stream));
    DevShard sh = dev_shard(h, c, d_export, cap);
    sh.by_dest = by_dest ? 1u : 0u;
    ev_begin(c, 1);
    launch_expand(c->stream, h->dev_graph(c), h->dev_frontier(*c), level, (uint8_t *)d_has, (uint8_t *)d_err, sh);
    ev_end(c);
    c->stats.expand_launches++;
    c->stats.levels_last = level;
    c->stats.check_passes += level == 1 ? 1 : 0;
    rc = shard_report(c, level, out);
    if (rc) return rc;
    if (by_dest) {
        uint64_t mx = 0;
        for (uint32_t d = 0; d < h->shard.world; d++) {
            by_dest_out[d] = c->h_status[2 * kLevelSlots + 2 + d];
            mx = std::max(mx, by_dest_out[d]);
        }
        out->exported = mx;  // the largest per-destination count: what the caller sizes a retry by
    }
    return ACL_OK;
}

}  // namespace

extern "C" {

int acl_shard_configure(acl_engine_t *h, uint32_t rank, uint32_t world) {
    if (world == 0 || rank >= world) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_configure: rank must be < world");
    std::lock_guard<RwLock> lk(h->state_mu);
    h->shard.rank = rank;
    h->shard.world = world;
    h->snap_valid = false;
    h->set_dev_valid(false);
    h->set_rev_uploaded(false);
    return ACL_OK;
}

int acl_shard_of_type(acl_engine_t *h, int type) {
    std::shared_lock<RwLock> lk(h->state_mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return -1;
    return (int)shard_of_type(sc.defs[type].name, h->shard.world);
}

void *acl_shard_stream(acl_engine_t *h) {
    ShardCall sc;
    if (sc.begin(h, false, false)) return nullptr;
    return (void *)sc.c->stream;
}

int acl_shard_grow_frontier(acl_engine_t *h) {
    ShardCall sc;
    int rc = sc.begin(h, false, false);
    if (rc) return rc;
    if (sc.c->frontier_entries >= (uint64_t)kMaxFrontierChunks * kChunk) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "frontier capacity exceeded");
    sc.c->stats.overflow_retries++;
    return alloc_frontier(h, sc.c, sc.c->frontier_entries * 4);
}

int acl_shard_check_begin(acl_engine_t *h, const void *d_items, size_t n, void *d_has, void *d_err) {
    if (n && (!d_items || !d_has || !d_err)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_begin: NULL buffer");
    if (n > 0xFFFFFFFFu) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_begin: batch too large");
    ShardCall sc;
    int rc = sc.begin(h, true, false);
    if (rc) return rc;
    PassCtx *c = sc.c;
    if ((uint64_t)n > c->frontier_entries) {
        rc = alloc_frontier(h, c, (uint64_t)n * 4);
        if (rc) return rc;
    }
    ev_begin(c, 0);
    launch_seed(c->stream, h->dev_graph(c), h->dev_frontier(*c), (const uint4 *)d_items, (uint32_t)n, (uint8_t *)d_has, (uint8_t *)d_err,
                dev_shard(h, c, nullptr, 0));  // also resets the status block
    ev_end(c);
    c->stats.check_items += n;
    return ACL_OK;
}

int acl_shard_check_step(acl_engine_t *h, uint32_t level, void *d_has, void *d_err, void *d_export, size_t export_cap, acl_shard_step_t *out) {
    return check_step(h, level, d_has, d_err, d_export, export_cap, false, out, nullptr);
}

int acl_shard_check_step_by_dest(acl_engine_t *h, uint32_t level, void *d_has, void *d_err, void *d_export, size_t cap_per_dest, acl_shard_step_t *out,
                                 uint64_t *exported_by_dest) {
    if (!exported_by_dest) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_step_by_dest: NULL counts");
    return check_step(h, level, d_has, d_err, d_export, cap_per_dest, true, out, exported_by_dest);
}

int acl_shard_check_import(acl_engine_t *h, uint32_t level, const void *d_entries, size_t n) {
    if (n && !d_entries) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_import: NULL buffer");
    ShardCall sc;
    int rc = sc.begin(h, false, false);
    if (rc) return rc;
    rc = shard_ready(h, level);
    if (rc) return rc;
    PassCtx *c = sc.c;
    ev_begin(c, 0);
    launch_import(c->stream, h->dev_graph(c), h->dev_frontier(*c), level, (const uint4 *)d_entries, (uint32_t)n, dev_shard(h, c, nullptr, 0));
    ev_end(c);
    return ACL_OK;
}

int acl_shard_check_finish(acl_engine_t *h, const void *d_has, const void *d_err, size_t n, void *d_perm_out, void *d_err_out) {
    if (n && (!d_has || !d_err || !d_perm_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_check_finish: NULL buffer");
    ShardCall sc;
    int rc = sc.begin(h, false, false);
    if (rc) return rc;
    PassCtx *c = sc.c;
    ev_begin(c, 0);
    launch_finalize(c->stream, (uint32_t)n, (const uint8_t *)d_has, (const uint8_t *)d_err, (uint8_t *)d_perm_out, (int32_t *)d_err_out);
    ev_end(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    ev_collect(c);
    return ACL_OK;
}

int acl_shard_lookup_begin(acl_engine_t *h, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n) {
    if (n && !sids) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_begin: NULL buffer");
    ShardCall scall;
    int rc = scall.begin(h, true, true);
    if (rc) return rc;
    PassCtx *c = scall.c;
    const Schema &sc = h->store.schema();
    if (rtype < 0 || rtype >= (int)sc.defs.size() || stype < 0 || stype >= (int)sc.defs.size() || perm < 0 ||
        perm >= (int)sc.defs[rtype].members.size() || srel >= (int)sc.defs[stype].members.size())
        return fail(ACL_ERR_FAILED_PRECONDITION, "lookup: unknown type, permission or subject relation");
    const size_t vwords = std::max<size_t>((size_t)((h->snap.visited_bits + 31) / 32), 1);
    if (n * vwords > ((size_t)1 << 30)) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "lookup batch too large for one pass (visited bitmaps > 4 GiB)");
    if (n > c->frontier_entries) {
        rc = alloc_frontier(h, c, n * 4);
        if (rc) return rc;
    }
    h->lk_target = (uint32_t)sc.slot(rtype, perm);
    h->lk_n = n;
    const uint32_t key = sc.subject_key(stype, srel < 0 ? kNoRelation : srel);
    HIP_TRY(c->d_visited.ensure(std::max<size_t>(n, 1) * vwords));
    HIP_TRY(hipMemsetAsync(c->d_visited.p, 0, std::max<size_t>(n, 1) * vwords * 4, c->stream));
    DevFrontier f = h->dev_frontier(*c);
    std::vector<uint4> seeds(n);
    for (size_t i = 0; i < n; i++) seeds[i] = make_uint4(sids[i], (uint32_t)i, key /* dist 0 */, 0);
    const size_t need_chunks = (n + kChunk - 1) / kChunk;
    std::vector<uint32_t> st(kStatusWords, 0), cc(std::max<size_t>(need_chunks, f.nwaves), 0);
    st[0] = need_chunks > f.nwaves ? (uint32_t)(need_chunks - f.nwaves) : 0u;
    st[kLevelSlots] = n ? 1 : 0;
    for (size_t k = 0; k < need_chunks; k++) cc[k] = (uint32_t)std::min<size_t>(kChunk, n - k * kChunk);
    HIP_TRY(hipMemcpyAsync(c->d_status.p, st.data(), st.size() * 4, hipMemcpyHostToDevice, c->stream));
    if (n) HIP_TRY(hipMemcpyAsync(f.buf[0], seeds.data(), n * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(f.counts[0], cc.data(), cc.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ACL_OK;
}

int acl_shard_lookup_step(acl_engine_t *h, uint32_t iter, int phase, void *d_export, size_t export_cap, acl_shard_step_t *out) {
    if (!out || (export_cap && !d_export)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_step: NULL buffer");
    if (phase != ACL_SHARD_VISIT && phase != ACL_SHARD_EXPAND) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_step: bad phase");
    ShardCall sc;
    int rc = sc.begin(h, false, false);
    if (rc) return rc;
    rc = shard_ready(h, iter);
    if (rc) return rc;
    if (!h->dev0().rev_uploaded) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_shard_lookup_step without acl_shard_lookup_begin");
    PassCtx *c = sc.c;
    const size_t vwords = std::max<size_t>((size_t)((h->snap.visited_bits + 31) / 32), 1);
    DevReverse r = h->dev_reverse(c, (uint32_t)vwords);
    HIP_TRY(hipMemsetAsync(c->d_status.p + 2 * kLevelSlots + 1, 0, sizeof(uint32_t), c->stream));
    ev_begin(c, 1);
    launch_rev_expand(c->stream, r, h->dev_frontier(*c), iter, phase == ACL_SHARD_VISIT ? REV_VISIT : REV_EXPAND, dev_shard(h, c, d_export, export_cap));
    ev_end(c);
    c->stats.expand_launches++;
    return shard_report(c, iter, out);
}

int acl_shard_lookup_import(acl_engine_t *h, uint32_t iter, const void *d_entries, size_t n) {
    if (n && !d_entries) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_shard_lookup_import: NULL buffer");
    ShardCall sc;
    int rc = sc.begin(h, false, false);
    if (rc) return rc;
    rc = shard_ready(h, iter);
    if (rc) return rc;
    if (!h->dev0().rev_uploaded) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_shard_lookup_import without acl_shard_lookup_begin");
    PassCtx *c = sc.c;
    DevReverse r = h->dev_reverse(c, 0);
    launch_rev_import(c->stream, r, h->dev_frontier(*c), iter, (const uint4 *)d_entries, (uint32_t)n);
    return ACL_OK;
}

int acl_shard_lookup_finish(acl_engine_t *h, void *d_bitmaps_out, size_t bitmap_words) {
    ShardCall sc;
    int rc = sc.begin(h, false, false);
    if (rc) return rc;
    if (!h->dev0().rev_uploaded) return fail(ACL_ERR_FAILED_PRECONDITION, "acl_shard_lookup_finish without acl_shard_lookup_begin");
    PassCtx *c = sc.c;
    const uint32_t nobj = h->store.objects(h->store.schema().slot_owner[h->lk_target].first).count();
    const size_t need = (nobj + 31) / 32;
    if (h->lk_n && (!d_bitmaps_out || bitmap_words < need))
        return fail(ACL_ERR_INVALID_ARGUMENT, "lookup: bitmap too small (" + std::to_string(need) + " words needed)");
    const size_t vwords = std::max<size_t>((size_t)((h->snap.visited_bits + 31) / 32), 1);
    const size_t woff = h->snap.slot_bit_base[h->lk_target] / 32;
    // only the ids the snapshot's bitmap slot covers can be marked; ids interned since have no relationship here (zeros)
    const size_t cw = std::min(need, ((size_t)h->snap.slot_nobjects[h->lk_target] + 31) / 32);
    // rows of the result: only the owner of the resource type ever marks them, other shards hand back zeros
    HIP_TRY(hipMemsetAsync(d_bitmaps_out, 0, h->lk_n * bitmap_words * 4, c->stream));
    if (cw)
        HIP_TRY(hipMemcpy2DAsync(d_bitmaps_out, bitmap_words * 4, c->d_visited.p + woff, vwords * 4, cw * 4, h->lk_n, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ACL_OK;
}

}  // extern "C"
