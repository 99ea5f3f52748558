// engine_callers.cpp -- the callers either side of the kernels: PostFilter keep mask, PreFilter bitmap test,
// Watch change feed, snapshot self-check hook, micro-batching front-end.
#include <tuple>

#include "engine_internal.hpp"

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
    std::vector<uint32_t> sids, bms;
    std::vector<uint64_t> cnts;
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
        // Checks of the batch: one device pass
        items.clear();
        for (acl_engine::Waiter *w : batch)
            if (w->kind == 0) items.push_back(w->item);
        perm.assign(items.size(), 0);
        err.assign(items.size(), 0);
        int rc = items.empty() ? ACL_OK : acl_check_bulk_ids(h, items.data(), items.size(), perm.data(), err.data());
        const std::string msg = rc ? acl_last_error() : "";
        // LookupResources of the batch: one batched reverse walk per (resource type, permission, subject class)
        std::vector<acl_engine::Waiter *> lks;
        for (acl_engine::Waiter *w : batch)
            if (w->kind == 1) lks.push_back(w);
        std::sort(lks.begin(), lks.end(), [](const acl_engine::Waiter *a, const acl_engine::Waiter *b) {
            return std::tie(a->lk_rtype, a->lk_perm, a->lk_stype, a->lk_srel, a->lk_words) < std::tie(b->lk_rtype, b->lk_perm, b->lk_stype, b->lk_srel, b->lk_words);
        });
        uint64_t walks = 0;
        for (size_t g0 = 0; g0 < lks.size();) {
            size_t g1 = g0 + 1;
            while (g1 < lks.size() && std::tie(lks[g1]->lk_rtype, lks[g1]->lk_perm, lks[g1]->lk_stype, lks[g1]->lk_srel, lks[g1]->lk_words) ==
                                          std::tie(lks[g0]->lk_rtype, lks[g0]->lk_perm, lks[g0]->lk_stype, lks[g0]->lk_srel, lks[g0]->lk_words))
                g1++;
            const size_t m = g1 - g0, words = lks[g0]->lk_words;
            sids.resize(m);
            bms.assign(m * words, 0);
            cnts.assign(m, 0);
            for (size_t i = 0; i < m; i++) sids[i] = lks[g0 + i]->lk_sid;
            const int lrc = acl_lookup_resources_batch(h, lks[g0]->lk_rtype, lks[g0]->lk_perm, lks[g0]->lk_stype, lks[g0]->lk_srel, sids.data(), m, bms.data(),
                                                       words, cnts.data());
            const std::string lmsg = lrc ? acl_last_error() : "";
            for (size_t i = 0; i < m; i++) {
                acl_engine::Waiter *w = lks[g0 + i];
                w->rc = lrc;
                w->msg = lmsg;
                if (!lrc) {
                    std::memcpy(w->lk_bitmap, bms.data() + i * words, words * sizeof(uint32_t));
                    w->lk_count = cnts[i];
                }
            }
            walks++;
            g0 = g1;
        }
        {
            std::lock_guard<std::mutex> lk(h->q_mu);
            size_t ci = 0;
            for (acl_engine::Waiter *w : batch) {
                if (w->kind == 0) {
                    w->rc = rc;
                    w->msg = msg;
                    w->perm = perm[ci];
                    w->err = err[ci];
                    ci++;
                }
                w->done = true;
                w->cv.notify_one();
            }
            if (!items.empty()) h->mb_batches++;
            h->mb_items += items.size();
            h->mb_lookup_walks += walks;
            h->mb_lookups += lks.size();
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

int acl_batcher_lookup_stats(acl_engine_t *h, uint64_t *walks, uint64_t *lookups) {
    std::lock_guard<std::mutex> lk(h->q_mu);
    if (walks) *walks = h->mb_lookup_walks;
    if (lookups) *lookups = h->mb_lookups;
    return ACL_OK;
}

// One LookupResources request (lookups.go:65; one per list request, issued from its own goroutine: responsefilterer.go:165).
// While the batcher runs, concurrent requests for the same (resource type, permission, subject class) share ONE batched
// reverse walk.  Blocks until answered; bitmap_out as for acl_lookup_resources.
int acl_lookup_one(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap_out,
                   size_t bitmap_words, uint64_t *count_out) {
    if (!bitmap_out && bitmap_words) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_lookup_one: NULL bitmap");
    acl_engine::Waiter w;
    w.kind = 1;
    int rc = resolve_lookup(h, rtype, perm, stype, sid, srel, &w.lk_rtype, &w.lk_perm, &w.lk_stype, &w.lk_srel, &w.lk_sid);
    if (rc) return rc;
    w.lk_bitmap = bitmap_out;
    w.lk_words = bitmap_words;
    {
        std::unique_lock<std::mutex> lk(h->q_mu);
        if (h->batcher_on && !h->batcher_stop) {
            h->queue.push_back(&w);
            if (h->queue.size() == 1 || h->queue.size() >= h->mb_max_items) h->q_cv.notify_one();
            w.cv.wait(lk, [&] { return w.done; });
            if (w.rc) return fail(w.rc, w.msg);
            if (count_out) *count_out = w.lk_count;
            return ACL_OK;
        }
    }
    return acl_lookup_resources_batch(h, w.lk_rtype, w.lk_perm, w.lk_stype, w.lk_srel, &w.lk_sid, 1, bitmap_out, bitmap_words, count_out);
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

