// engine.cpp -- C ABI of libaclgpu.so (include/aclgpu.h): device memory, snapshot upload,
// the level loop around the frontier kernels, string <-> id plumbing.
//
// Reference behaviour mirrored at this boundary:
//   CheckBulkPermissions : pairs are index-aligned with items (pkg/authz/check.go:54-57),
//                          per-item error or permissionship (check.go:55-63)
//   LookupResources      : set of ids with HAS_PERMISSION, order irrelevant (lookups.go:85-88,129)
//   every read is fully consistent (check.go:41-46): a write is visible to the next call.
// There is no CPU evaluation path: without a GPU acl_open() fails.
#include "engine_internal.hpp"

namespace aclint {

thread_local std::string g_last_error;


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

}  // namespace aclint

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

extern "C++" {
namespace aclint {
int32_t intern_check_item(acl_engine_t *h, const acl_check_item_t &it, acl_item_t *out) {
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
}  // namespace aclint
}  // extern "C++"

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

extern "C++" {
namespace aclint {
// LookupResourcesRequest strings -> ids (lookups.go:49-62); the subject is interned so `stype:sid#srel` can be its own member
int resolve_lookup(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, int *rt_out, int *pm_out,
                   int *st_out, int *sr_out, uint32_t *sub_out) {
    std::lock_guard<std::mutex> lk(h->mu);
    std::unique_lock<std::shared_mutex> nlk(h->names_mu);
    if (empty(rtype) || empty(perm) || empty(stype) || empty(sid)) return fail(ACL_ERR_INVALID_ARGUMENT, "invalid LookupResourcesRequest: empty field");
    if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
    const Schema &sc = h->store.schema();
    int sr = -1;
    const int rt = sc.type_of(rtype);
    if (rt < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("object definition `") + rtype + "` not found");
    const int pm = sc.defs[rt].find(perm);
    if (pm < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("relation/permission `") + perm + "` not found under definition `" + rtype + "`");
    const int st = sc.type_of(stype);
    if (st < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("object definition `") + stype + "` not found");
    if (!empty(srel) && std::strcmp(srel, "...") != 0) {
        sr = sc.defs[st].find(srel);
        if (sr < 0) return fail(ACL_ERR_FAILED_PRECONDITION, std::string("relation `") + srel + "` not found under definition `" + stype + "`");
    }
    *sub_out = h->store.objects(st).intern(sid);
    // a subject the reverse rows have no room for (beyond the headroom ids): they are rebuilt by this lookup
    if (sr >= 0 && h->snap.has_reverse && h->store.objects(st).count() > h->snap.slot_nobjects[sc.slot(st, sr)]) {
        h->rev_uploaded = false;
        h->snap.has_reverse = false;
    }
    *rt_out = rt;
    *pm_out = pm;
    *st_out = st;
    *sr_out = sr;
    return ACL_OK;
}
}  // namespace aclint
}  // extern "C++"

int acl_lookup_resources(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap,
                         size_t words, uint64_t *count) {
    int rt, pm, st, sr;
    uint32_t sub;
    int rc = resolve_lookup(h, rtype, perm, stype, sid, srel, &rt, &pm, &st, &sr, &sub);
    if (rc) return rc;
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
