// engine_callers.cpp -- the callers either side of the kernels: PostFilter keep mask, PreFilter bitmap test,
// Watch change feed, snapshot self-check hook, micro-batching front-end.
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <sched.h>

#include <climits>
#include <tuple>

#include "engine_internal.hpp"

// ---------------------------------------------------------------- micro-batching front-end
// The proxy issues many concurrent 1-item checks (check.go:76-94: one goroutine per check expression; watch.go:50: one
// per update) and one LookupResources per list request (responsefilterer.go:165).  Callers append to the OPEN batch (a
// few dozen nanoseconds under one mutex) and sleep on that batch's own futex word; dispatcher threads -- one per
// evaluation context, so passes overlap on the device -- close the open batch, answer it with one device pass and wake
// exactly its callers with ONE futex call.  (Round 1 woke every caller through its own condition variable while holding
// the queue lock: 263 k checks/s at 64 threads, collapsing to 46 k/s at 1 024.)
namespace {

inline long futex(std::atomic<uint32_t> *addr, int op, uint32_t val, const timespec *ts) {
    return syscall(SYS_futex, reinterpret_cast<uint32_t *>(addr), op, val, ts, nullptr, 0);
}

struct LookupReq {
    int rtype, perm, stype, srel;
    uint32_t sid;
    size_t words;
    // filled by the dispatcher
    int rc = 0;
    std::string msg;
    std::vector<uint32_t> bitmap;
    uint64_t count = 0;
};

struct Batch {
    std::atomic<uint32_t> done{0};
    std::atomic<uint32_t> refs{1};  // the queue's own reference + one per caller
    std::vector<acl_item_t> items;
    std::vector<uint8_t> perm;
    std::vector<int32_t> err;
    std::vector<LookupReq> lookups;
    int rc = 0;
    std::string msg;
    int64_t opened_ns = 0;
    void unref() {
        if (refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete this;
    }
};

}  // namespace

struct acl_engine::Batcher {
    std::mutex mu;
    std::condition_variable cv;  // dispatchers only
    Batch *open = nullptr;
    uint32_t idle = 0;      // dispatchers parked on cv
    uint32_t in_flight = 0;  // passes on the device
    bool stop = false;
    std::vector<std::thread> threads;
    uint32_t max_items = 4096, wait_us = 200;
    std::atomic<uint64_t> batches{0}, items{0}, lookup_walks{0}, lookups{0};
    std::atomic<uint32_t> sleepers{0};
    unsigned cores = 1;
};

namespace {

void answer_batch(acl_engine_t *h, Batch *b) {
    if (!b->items.empty()) {
        b->perm.assign(b->items.size(), 0);
        b->err.assign(b->items.size(), 0);
        b->rc = acl_check_bulk_ids(h, b->items.data(), b->items.size(), b->perm.data(), b->err.data());
        if (b->rc) b->msg = acl_last_error();
    }
    // LookupResources of the batch: one batched reverse walk per (resource type, permission, subject class)
    uint64_t walks = 0;
    if (!b->lookups.empty()) {
        std::vector<LookupReq *> lks;
        for (LookupReq &l : b->lookups) lks.push_back(&l);
        auto key = [](const LookupReq *a) { return std::tie(a->rtype, a->perm, a->stype, a->srel, a->words); };
        std::stable_sort(lks.begin(), lks.end(), [&](const LookupReq *a, const LookupReq *c) { return key(a) < key(c); });
        std::vector<uint32_t> sids, bms;
        std::vector<uint64_t> cnts;
        for (size_t g0 = 0; g0 < lks.size();) {
            size_t g1 = g0 + 1;
            while (g1 < lks.size() && key(lks[g1]) == key(lks[g0])) g1++;
            const size_t m = g1 - g0, words = lks[g0]->words;
            sids.resize(m);
            bms.assign(m * std::max<size_t>(words, 1), 0);
            cnts.assign(m, 0);
            for (size_t i = 0; i < m; i++) sids[i] = lks[g0 + i]->sid;
            const int lrc = acl_lookup_resources_batch(h, lks[g0]->rtype, lks[g0]->perm, lks[g0]->stype, lks[g0]->srel, sids.data(), m, bms.data(), words, cnts.data());
            const std::string lmsg = lrc ? acl_last_error() : "";
            for (size_t i = 0; i < m; i++) {
                LookupReq *w = lks[g0 + i];
                w->rc = lrc;
                w->msg = lmsg;
                if (!lrc) {
                    w->bitmap.assign(bms.begin() + (long)(i * words), bms.begin() + (long)((i + 1) * words));
                    w->count = cnts[i];
                }
            }
            walks++;
            g0 = g1;
        }
    }
    acl_engine::Batcher &B = *h->batcher;
    if (!b->items.empty()) B.batches++;
    B.items += b->items.size();
    B.lookup_walks += walks;
    B.lookups += b->lookups.size();
    b->done.store(1, std::memory_order_release);
    futex(&b->done, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr);  // exactly this batch's callers, one system call
    b->unref();
}

void dispatcher_loop(acl_engine_t *h) {
    acl_engine::Batcher &B = *h->batcher;
    for (;;) {
        Batch *b = nullptr;
        {
            std::unique_lock<std::mutex> lk(B.mu);
            B.idle++;
            B.cv.wait(lk, [&] { return B.stop || (B.open && (!B.open->items.empty() || !B.open->lookups.empty())); });
            B.idle--;
            if (!B.open || (B.open->items.empty() && B.open->lookups.empty())) {
                if (B.stop) return;
                continue;
            }
            // idle device and a small batch: let concurrent callers pile on for at most wait_us.  When a pass is already in
            // flight, that pass WAS the batching window: whoever arrived during it goes now.
            if (!B.stop && B.in_flight == 0 && B.wait_us && B.open->items.size() + B.open->lookups.size() < B.max_items) {
                const int64_t until = B.open->opened_ns + (int64_t)B.wait_us * 1000;
                while (!B.stop && B.open && B.open->items.size() + B.open->lookups.size() < B.max_items) {
                    const int64_t now = mono_ns();
                    if (now >= until) break;
                    B.cv.wait_for(lk, std::chrono::nanoseconds(until - now));
                }
                if (!B.open) continue;  // another dispatcher took it meanwhile
            }
            b = B.open;
            B.open = nullptr;
            B.in_flight++;
        }
        answer_batch(h, b);
        {
            std::lock_guard<std::mutex> lk(B.mu);
            B.in_flight--;
        }
    }
}

// appends to the open batch; returns the batch (one reference for the caller) and the caller's index in it
Batch *enqueue(acl_engine_t *h, const acl_item_t *item, const LookupReq *lk, size_t *index) {
    acl_engine::Batcher *B = h->batcher;
    if (!B) return nullptr;
    std::unique_lock<std::mutex> g(B->mu);
    if (B->stop || B->threads.empty()) return nullptr;  // no batcher running
    if (!B->open) {
        B->open = new Batch();
        B->open->opened_ns = mono_ns();
    }
    Batch *b = B->open;
    const bool was_empty = b->items.empty() && b->lookups.empty();
    if (item) {
        *index = b->items.size();
        b->items.push_back(*item);
    } else {
        *index = b->lookups.size();
        b->lookups.push_back(*lk);
    }
    b->refs.fetch_add(1, std::memory_order_relaxed);
    const bool full = b->items.size() + b->lookups.size() >= B->max_items;
    const bool wake = (was_empty || full) && B->idle > 0;
    g.unlock();
    if (wake) B->cv.notify_one();
    return b;
}

// parks the caller until its batch is answered: a short spin first when cores are to spare (a pass takes tens of
// microseconds, a futex sleep + wake about as long), then the batch's futex
int await_batch(acl_engine_t *h, Batch *b, const CallOpts &opts) {
    acl_engine::Batcher &B = *h->batcher;
    const bool watched = opts.cancel || opts.deadline_ns;
    if ((B.sleepers.load(std::memory_order_relaxed) + 4) * 2 < B.cores) {  // (+ the dispatchers, which spin in their stream syncs)
        const int64_t spin_until = mono_ns() + 30000;
        B.sleepers.fetch_add(1, std::memory_order_relaxed);
        while (!b->done.load(std::memory_order_acquire) && mono_ns() < spin_until) {
            for (int i = 0; i < 32; i++) __builtin_ia32_pause();
        }
        B.sleepers.fetch_sub(1, std::memory_order_relaxed);
    }
    if (!b->done.load(std::memory_order_acquire)) {
        B.sleepers.fetch_add(1, std::memory_order_relaxed);
        while (!b->done.load(std::memory_order_acquire)) {
            if (watched) {
                int rc = check_opts(opts);
                if (rc) {
                    B.sleepers.fetch_sub(1, std::memory_order_relaxed);
                    return rc;  // the batch still answers the abandoned slot; nobody reads it
                }
                timespec ts{0, 500000};
                futex(&b->done, FUTEX_WAIT_PRIVATE, 0, &ts);
            } else {
                futex(&b->done, FUTEX_WAIT_PRIVATE, 0, nullptr);
            }
        }
        B.sleepers.fetch_sub(1, std::memory_order_relaxed);
    }
    return ACL_OK;
}

// hardware threads this process may actually use: affinity mask, capped by the cgroup CPU quota (a container on a
// 256-thread box may be allowed 16 cores' worth of time: spinning there burns the quota every thread shares)
unsigned usable_cores() {
    unsigned c = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) c = std::min<unsigned>(c, (unsigned)std::max(1, CPU_COUNT(&set)));
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long per = 0;
        if (std::fscanf(f, "%31s %ld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) c = std::min<unsigned>(c, (unsigned)std::max(1L, (std::atol(q) + per / 2) / per));
        std::fclose(f);
    }
    return c;
}

CallOpts to_opts(const acl_call_opts_t *o) {
    CallOpts opts;
    if (o) {
        opts.cancel = o->cancel;
        if (o->timeout_ns > 0) opts.deadline_ns = mono_ns() + o->timeout_ns;
    }
    return opts;
}

}  // namespace

namespace aclint {

void batcher_create(acl_engine_t *h) { h->batcher = new acl_engine::Batcher(); }
void batcher_destroy(acl_engine_t *h) {
    if (h->batcher && h->batcher->open) h->batcher->open->unref();
    delete h->batcher;
    h->batcher = nullptr;
}

// one LookupResources with interned arguments: rides the micro-batcher when it runs, else a walk of its own
int lookup_one_routed(acl_engine_t *h, int rt, int pm, int st, int sr, uint32_t sub, uint32_t *bitmap_out, size_t words, uint64_t *count_out,
                      const CallOpts &opts) {
    LookupReq lk{rt, pm, st, sr, sub, words};
    size_t idx = 0;
    Batch *b = enqueue(h, nullptr, &lk, &idx);
    if (!b) return lookup_batch_call(h, rt, pm, st, sr, &sub, 1, bitmap_out, words, count_out, opts);
    int rc = await_batch(h, b, opts);
    if (rc == ACL_OK) {
        LookupReq &r = b->lookups[idx];
        if (r.rc) rc = fail(r.rc, r.msg);
        else {
            if (words) std::memcpy(bitmap_out, r.bitmap.data(), words * sizeof(uint32_t));
            if (count_out) *count_out = r.count;
        }
    }
    b->unref();
    return rc;
}

}  // namespace aclint

// ---------------------------------------------------------------- callers either side of the kernels
extern "C" {

// filterItemsWithBulkPermissions (postfilter.go:58-182) fused: ONE bulk check of the K*F resolved pairs and the
// per-list-item AND, on the device; only K bytes come back.
static int keep_device(acl_engine_t *h, PassCtx *c, const void *d_items, size_t n, const void *d_item_off, size_t k_items, void *d_keep_out) {
    HIP_TRY(c->d_perm.ensure(std::max<size_t>(n, 1)));
    int rc = check_device(h, c, (const uint4 *)d_items, n, c->d_perm.p, nullptr);
    if (rc) return rc;
    launch_keep(c->stream, (uint32_t)k_items, (const uint32_t *)d_item_off, c->d_perm.p, (uint8_t *)d_keep_out);
    return ACL_OK;
}

int acl_check_bulk_keep_ids_device(acl_engine_t *h, const void *d_items, size_t n, const void *d_item_off, size_t k_items, void *d_keep_out) {
    if ((n && !d_items) || (k_items && (!d_item_off || !d_keep_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_ids_device: NULL buffer");
    Eval ev;
    int rc = ev.begin(h, false);
    if (rc) return rc;
    rc = keep_device(h, ev.c, d_items, n, d_item_off, k_items, d_keep_out);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ev.c->stream));
    ev_collect(ev.c);
    return ACL_OK;
}

int acl_check_bulk_keep_ids(acl_engine_t *h, const acl_item_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out) {
    if ((n && !items) || (k_items && (!item_off || !keep_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_ids: NULL buffer");
    if (!k_items) return ACL_OK;
    for (size_t i = 0; i < k_items; i++)
        if (item_off[i] > item_off[i + 1] || item_off[i + 1] > n) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_keep_ids: item_off must ascend and end within n");
    Eval ev;
    int rc = ev.begin(h, false);
    if (rc) return rc;
    PassCtx *c = ev.c;
    HIP_TRY(c->d_items.ensure(std::max<size_t>(n, 1)));
    HIP_TRY(c->d_itemoff.ensure(k_items + 1));
    HIP_TRY(c->d_keep.ensure(k_items));
    HIP_TRY(c->h_in.ensure(n * sizeof(acl_item_t) + (k_items + 1) * sizeof(uint32_t)));
    std::memcpy(c->h_in.p, items, n * sizeof(acl_item_t));
    uint32_t *h_off = (uint32_t *)((char *)c->h_in.p + n * sizeof(acl_item_t));
    std::memcpy(h_off, item_off, (k_items + 1) * sizeof(uint32_t));
    if (n) HIP_TRY(hipMemcpyAsync(c->d_items.p, c->h_in.p, n * sizeof(acl_item_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_itemoff.p, h_off, (k_items + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    rc = keep_device(h, c, c->d_items.p, n, c->d_itemoff.p, k_items, c->d_keep.p);
    if (rc) return rc;
    HIP_TRY(c->h_out.ensure(k_items));
    HIP_TRY(hipMemcpyAsync(c->h_out.p, c->d_keep.p, k_items, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ev_collect(c);
    std::memcpy(keep_out, c->h_out.p, k_items);
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
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
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
    if (ntypes < 0 || (ntypes && !types)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_watch_poll: bad argument");
    std::shared_lock<RwLock> lk(h->state_mu);
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
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
    std::lock_guard<RwLock> lk(h->state_mu);
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

int acl_batcher_start(acl_engine_t *h, uint32_t max_items, uint32_t max_wait_us) {
    std::lock_guard<std::mutex> lk(h->batcher_mu);
    acl_engine::Batcher &B = *h->batcher;
    std::lock_guard<std::mutex> g(B.mu);
    if (!B.threads.empty()) return fail(ACL_ERR_FAILED_PRECONDITION, "batcher already running");
    B.max_items = max_items ? max_items : 4096;
    B.wait_us = max_wait_us;
    B.cores = usable_cores();
    B.stop = false;
    // one dispatcher per evaluation context the engine may open, but no more than a quarter of the usable cores (a
    // dispatcher's stream synchronisation spins); store-only engines: one, it only reports the error
    const uint32_t nd = h->store_only ? 1u : std::max<uint32_t>(1, std::min<uint32_t>({h->max_ctx, 4u, std::max(1u, B.cores / 4)}));
    for (uint32_t i = 0; i < nd; i++) B.threads.emplace_back(dispatcher_loop, h);
    return ACL_OK;
}

int acl_batcher_stop(acl_engine_t *h) {
    std::lock_guard<std::mutex> lk(h->batcher_mu);
    if (!h->batcher) return ACL_OK;
    acl_engine::Batcher &B = *h->batcher;
    std::vector<std::thread> threads;
    {
        std::lock_guard<std::mutex> g(B.mu);
        if (B.threads.empty()) return ACL_OK;
        B.stop = true;  // enqueue() refuses from here on: callers fall back to passes of their own
        threads.swap(B.threads);
    }
    B.cv.notify_all();
    for (auto &t : threads) t.join();  // dispatchers leave only when nothing is queued
    return ACL_OK;
}

int acl_batcher_stats(acl_engine_t *h, uint64_t *batches, uint64_t *items) {
    std::lock_guard<std::mutex> lk(h->batcher_mu);
    if (batches) *batches = h->batcher->batches.load();
    if (items) *items = h->batcher->items.load();
    return ACL_OK;
}

int acl_batcher_lookup_stats(acl_engine_t *h, uint64_t *walks, uint64_t *lookups) {
    std::lock_guard<std::mutex> lk(h->batcher_mu);
    if (walks) *walks = h->batcher->lookup_walks.load();
    if (lookups) *lookups = h->batcher->lookups.load();
    return ACL_OK;
}

// One LookupResources request (lookups.go:65; one per list request, issued from its own goroutine: responsefilterer.go:165).
// While the batcher runs, concurrent requests for the same (resource type, permission, subject class) share ONE batched
// reverse walk.  Blocks until answered, cancelled or timed out; bitmap_out as for acl_lookup_resources.
int acl_lookup_one_opts(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap_out,
                        size_t bitmap_words, uint64_t *count_out, const acl_call_opts_t *o) {
    if (!bitmap_out && bitmap_words) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_lookup_one: NULL bitmap");
    const CallOpts opts = to_opts(o);
    int rt, pm, st, sr;
    uint32_t sub;
    int rc = resolve_lookup(h, rtype, perm, stype, sid, srel, &rt, &pm, &st, &sr, &sub);
    if (rc) return rc;
    return lookup_one_routed(h, rt, pm, st, sr, sub, bitmap_out, bitmap_words, count_out, opts);
}
int acl_lookup_one(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap_out,
                   size_t bitmap_words, uint64_t *count_out) {
    return acl_lookup_one_opts(h, rtype, perm, stype, sid, srel, bitmap_out, bitmap_words, count_out, nullptr);
}

// CheckPermission (watch.go:50) / a 1-item CheckBulkPermissions (check.go:23-48).  Blocks until answered.
int acl_check_one_opts(acl_engine_t *h, const acl_check_item_t *item, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *o) {
    if (!item || !perm_out || !err_out) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_one: NULL argument");
    acl_item_t it;
    {
        std::shared_lock<std::shared_mutex> nlk(h->names_mu);  // string -> id reads only: callers do not serialise on the engine
        if (!h->store.has_schema()) return fail(ACL_ERR_FAILED_PRECONDITION, "no schema loaded");
        *perm_out = ACL_PERM_UNSPECIFIED;
        *err_out = intern_check_item(h, *item, &it);
        if (*err_out) return ACL_OK;
    }
    const CallOpts opts = to_opts(o);
    size_t idx = 0;
    Batch *b = enqueue(h, &it, nullptr, &idx);
    if (!b) return acl_check_bulk_ids_opts(h, &it, 1, perm_out, err_out, o);  // no batcher: a device pass of its own
    int rc = await_batch(h, b, opts);
    if (rc == ACL_OK) {
        if (b->rc) rc = fail(b->rc, b->msg);
        else {
            *perm_out = b->perm[idx];
            *err_out = b->err[idx];
        }
    }
    b->unref();
    return rc;
}
int acl_check_one(acl_engine_t *h, const acl_check_item_t *item, uint8_t *perm_out, int32_t *err_out) {
    return acl_check_one_opts(h, item, perm_out, err_out, nullptr);
}

}  // extern "C"
