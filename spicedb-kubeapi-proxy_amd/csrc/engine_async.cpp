// engine_async.cpp -- pipelined host-buffer Check (acl_check_bulk_ids_submit / acl_ticket_wait) and pinned host buffers.
//
// SURVEY.md 8(d) defines batch throughput on the ABI call that takes HOST ids: H2D + kernels + D2H.  One such call
// leaves the device idle during its copies and the copy engines idle during its kernels; submit/wait runs several of
// them at once, each on its own evaluation context (own HIP stream), so the H2D of batch N+1 and the D2H of batch N-1
// overlap the kernels of batch N.  A Go caller gets the same overlap from goroutines blocking in acl_check_bulk_ids;
// this form is for single-threaded hosts (bench.py) and for callers that want to keep a window of batches in flight.
#include "engine_internal.hpp"

// ACL_TRACE_PIPELINE=1: every hand-off of the submit / wait pipeline with its time, printed when the pool shuts down (debugging aid)
namespace {
struct PipeTrace {
    bool on = getenv("ACL_TRACE_PIPELINE") != nullptr;
    std::mutex mu;
    std::vector<std::tuple<uint64_t, const char *, int64_t>> ev;
    std::atomic<uint64_t> seq{0};
    void mark(uint64_t id, const char *what) {
        if (!on) return;
        const int64_t t = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        std::lock_guard<std::mutex> lk(mu);
        ev.emplace_back(id, what, t);
    }
    void dump() {
        if (!on || ev.empty()) return;
        const int64_t t0 = std::get<2>(ev.front());
        for (auto &e : ev) std::fprintf(stderr, "aclgpu-pipeline %llu %s %.1f\n", (unsigned long long)std::get<0>(e), std::get<1>(e), (std::get<2>(e) - t0) / 1e3);
        ev.clear();
    }
} g_trace;
}  // namespace

struct acl_ticket {
    uint64_t seq = 0;
    const acl_item_t *items = nullptr;
    size_t n = 0;
    uint8_t *perm = nullptr;
    int32_t *err = nullptr;
    // pipelined (large) batches: the evaluation context is held from submit to wait
    bool staged_pipeline = false;
    bool chained = false;  // its kernel was enqueued behind the previous batch's (chained_enqueue): the waiter completes the pass
    Eval ev;
    uint8_t *hp = nullptr;  // where the D2H copies land (the caller's pinned buffers or the context's staging)
    int32_t *he = nullptr;
    int rc = 0;
    std::string msg;
    bool done = false;
    std::mutex mu;
    std::condition_variable cv;
};

struct AsyncPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<acl_ticket *> queue;     // small batches: any worker runs the whole call
    std::deque<acl_ticket *> compute;   // chip-filling batches: ONE worker runs their kernels, one batch at a time, in order
    std::deque<acl_ticket *> completing;  // ... enqueued on the device: the completer waits for each in turn, finishes the pass and gives lock + context back
    std::condition_variable ccv;
    std::vector<std::thread> workers;
    std::thread compute_worker, completer;
    bool stop = false, compute_done = false;
};

namespace {

void finish(acl_ticket *t, int rc) {
    std::string msg = rc ? acl_last_error() : "";
    // (notified under the lock: the waiter deletes the ticket as soon as it has seen `done`, so nothing of *t may be touched
    //  once the lock is gone)
    std::lock_guard<std::mutex> lk(t->mu);
    t->rc = rc;
    t->msg = std::move(msg);
    t->done = true;
    t->cv.notify_one();
}

void worker_loop(acl_engine_t *h, AsyncPool *P) {
    for (;;) {
        acl_ticket *t = nullptr;
        {
            std::unique_lock<std::mutex> lk(P->mu);
            P->cv.wait(lk, [&] { return P->stop || !P->queue.empty(); });
            if (P->queue.empty()) return;
            t = P->queue.front();
            P->queue.pop_front();
        }
        finish(t, acl_check_bulk_ids(h, t->items, t->n, t->perm, t->err));
    }
}

// The pipeline of chip-filling batches, owned by ONE worker: it takes a context for a batch and starts its H2D copy as early as
// a context is free (up to two batches ahead of the one whose kernels run), runs the batches' kernels strictly one after the
// other (a batch this size fills every wave slot: two at once only take turns), and enqueues each batch's D2H copies without
// waiting for them -- they drain under the next batch's kernels.  The waiter synchronises the batch's stream, copies
// out of staging if the caller's buffers are not pinned, and gives the context back.
int stage_copies(acl_engine_t *h, acl_ticket *t);
int stage(acl_engine_t *h, acl_ticket *t, bool may_block) {
    g_trace.mark(t->seq, may_block ? "stage_begin_blocking" : "stage_begin");
    int rc = t->ev.begin(h, false, CallOpts(), -1, !may_block, chains(h, t->n));
    if (rc) return rc;
    g_trace.mark(t->seq, "context");
    rc = stage_copies(h, t);
    g_trace.mark(t->seq, "h2d_enqueued");
    if (rc) t->ev.end();  // (a batch that failed to stage kept the shared state lock and its context until the caller waited)
    return rc;
}
int stage_copies(acl_engine_t *h, acl_ticket *t) {
    PassCtx *c = t->ev.c;
    const size_t n = t->n;
    HIP_TRY(c->d_items.ensure(n));
    HIP_TRY(c->d_perm.ensure(n));
    HIP_TRY(c->d_errout.ensure(n));
    g_trace.mark(t->seq, "ensured");
    const void *src = t->items;
    if (!h->is_pinned(t->items, n * sizeof(acl_item_t))) {
        HIP_TRY(c->h_in.ensure(n * sizeof(acl_item_t)));
        std::memcpy(c->h_in.p, t->items, n * sizeof(acl_item_t));
        src = c->h_in.p;
    }
    const bool pin_p = h->is_pinned(t->perm, n), pin_e = !t->err || h->is_pinned(t->err, n * sizeof(int32_t));
    t->hp = t->perm;
    t->he = t->err;
    if (!pin_p || !pin_e) {
        HIP_TRY(c->h_out.ensure(n * 5 + 64));
        if (!pin_e) t->he = (int32_t *)c->h_out.p;
        if (!pin_p) t->hp = (uint8_t *)c->h_out.p + n * 4;
    }
    g_trace.mark(t->seq, "before_h2d");
    HIP_TRY(hipMemcpyAsync(c->d_items.p, src, n * sizeof(acl_item_t), hipMemcpyHostToDevice, c->stream));
    t->staged_pipeline = true;
    return ACL_OK;
}

void compute_loop(acl_engine_t *h, AsyncPool *P) {
    (void)hipSetDevice(h->dev0().device);  // (Eval::begin moves this thread to the device of whatever context it takes)
    std::deque<acl_ticket *> staged;  // context taken, H2D under way
    for (;;) {
        // look ahead: stage queued batches while contexts are free (never waiting for one)
        for (;;) {
            acl_ticket *t = nullptr;
            {
                std::lock_guard<std::mutex> lk(P->mu);
                if (staged.size() >= 3 || P->compute.empty()) break;
                t = P->compute.front();
            }
            const int rc = stage(h, t, staged.empty());  // with nothing staged there is nothing else to do: wait for a context
            if (rc == kNoContextFree) break;
            {
                std::lock_guard<std::mutex> lk(P->mu);
                P->compute.pop_front();
            }
            if (rc) finish(t, rc);
            else staged.push_back(t);
        }
        if (staged.empty()) {
            std::unique_lock<std::mutex> lk(P->mu);
            P->cv.wait(lk, [&] { return P->stop || !P->compute.empty(); });
            if (P->compute.empty()) return;
            continue;
        }
        acl_ticket *t = staged.front();
        staged.pop_front();
        PassCtx *c = t->ev.c;
        // The kernel goes behind the previous batch's ON THE DEVICE (an event between the two contexts' streams) and this thread moves on
        // to the next batch without waiting for it: launching batch N + 1 only after synchronising batch N left the chip idle for a
        // wake-up and a launch between two kernels.  The waiter synchronises, and redoes the pass on the level loop if a block overflowed.
        // (Windows of 3-4 tickets once measured 630 M/s against 870 M/s at 2 and 6: not the pipeline -- the HIP runtime sets up its copy paths lazily, the
        //  first hipMemcpyAsync that finds two other copies in flight blocks ~7 ms, once per process, and a short warm-up left that for the timed region:
        //  profiles/r03_submit_window_trace.txt.  Copies issued at open did not pre-empt it.)
        int rc = chained_enqueue(h, c, t->n);
        g_trace.mark(t->seq, "kernel_enqueued");
        if (rc == ACL_OK) {
            t->chained = true;
        } else if (rc == kChainDeclined) {
            std::lock_guard<std::mutex> tk(c->dev->compute_mu);  // (blocking callers with chip-filling batches take turns with the pipeline)
            rc = check_device(h, c, c->d_items.p, t->n, c->d_perm.p, c->d_errout.p);  // (the stream already carries the H2D)
        }
        if (!rc) {
            hipError_t e = hipMemcpyAsync(t->hp, c->d_perm.p, t->n, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && t->err) e = hipMemcpyAsync(t->he, c->d_errout.p, t->n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(ACL_ERR_INTERNAL, std::string("result copy: ") + hipGetErrorString(e));
        }
        if (rc) {
            (void)hipStreamSynchronize(c->stream);
            t->ev.end();
            finish(t, rc);
            continue;
        }
        g_trace.mark(t->seq, "d2h_enqueued");
        {  // everything is enqueued: the completer takes it from here, this thread moves on to the next batch
            std::lock_guard<std::mutex> lk(P->mu);
            P->completing.push_back(t);
        }
        P->ccv.notify_one();
    }
}

// Completes pipelined batches in the order their kernels were enqueued: waits for the batch's stream, redoes the pass on the level loop if a block
// of the walk ran out of private frontier, copies out of staging when the caller's buffers are not pinned, and gives the shared state lock and
// the context back -- WITHOUT the caller: a ticket nobody waits for no longer pins the engine's state lock (a writer used to starve behind
// it) or a context, and acl_ticket_wait is a plain wait for the ticket's answer.
void completer_loop(acl_engine_t *h, AsyncPool *P) {
    (void)hipSetDevice(h->dev0().device);
    for (;;) {
        acl_ticket *t = nullptr;
        {
            std::unique_lock<std::mutex> lk(P->mu);
            P->ccv.wait(lk, [&] { return !P->completing.empty() || P->compute_done; });
            if (P->completing.empty()) return;
            t = P->completing.front();
            P->completing.pop_front();
        }
        PassCtx *c = t->ev.c;
        int rc = ACL_OK;
        hipError_t e = hipSuccess;
        (void)hipSetDevice(c->dev->device);  // (the batch may have run on any replica)
        g_trace.mark(t->seq, "completer_takes");
        if (t->chained) {
            rc = chained_finish(h, c, t->n);  // synchronises the stream (kernel + result copies)
            g_trace.mark(t->seq, "stream_synchronised");
            if (rc == kChainDeclined) {       // a block ran out of private frontier: the level loop, and the copies once more
                {
                    std::lock_guard<std::mutex> tk(c->dev->compute_mu);
                    rc = check_device(h, c, c->d_items.p, t->n, c->d_perm.p, c->d_errout.p, false);
                }
                if (!rc) {
                    e = hipMemcpyAsync(t->hp, c->d_perm.p, t->n, hipMemcpyDeviceToHost, c->stream);
                    if (e == hipSuccess && t->err) e = hipMemcpyAsync(t->he, c->d_errout.p, t->n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                }
            }
        } else {
            e = hipStreamSynchronize(c->stream);  // the D2H copies (everything else on this stream finished before them)
            ev_collect(c);
        }
        if (!rc && e != hipSuccess) rc = fail(ACL_ERR_INTERNAL, std::string("result copy: ") + hipGetErrorString(e));
        if (!rc) {
            if (t->hp != t->perm) std::memcpy(t->perm, t->hp, t->n);
            if (t->err && t->he != t->err) std::memcpy(t->err, t->he, t->n * sizeof(int32_t));
        }
        t->ev.end();
        g_trace.mark(t->seq, "finished");
        finish(t, rc);
    }
}

}  // namespace

namespace aclint {

void async_shutdown(acl_engine_t *h) {
    AsyncPool *P = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->async_mu);
        P = h->async;
        h->async = nullptr;
    }
    if (!P) return;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->stop = true;
    }
    P->cv.notify_all();
    for (auto &t : P->workers) t.join();  // drains what is queued first
    if (P->compute_worker.joinable()) P->compute_worker.join();
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->compute_done = true;  // (nothing more will be handed to the completer)
    }
    P->ccv.notify_all();
    if (P->completer.joinable()) P->completer.join();
    delete P;
    g_trace.dump();
}

}  // namespace aclint

extern "C" {

int acl_check_bulk_ids_submit(acl_engine_t *h, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, acl_ticket_t **ticket_out) {
    if (!ticket_out || (n && (!items || !perm_out))) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_check_bulk_ids_submit: NULL argument");
    *ticket_out = nullptr;
    if (h->store_only) return fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU): Check / LookupResources are unavailable");
    AsyncPool *P;
    {
        std::lock_guard<std::mutex> lk(h->async_mu);
        if (!h->async) {
            h->async = new AsyncPool();
            for (uint32_t i = 0; i < std::max<uint32_t>(1, h->max_ctx); i++) h->async->workers.emplace_back(worker_loop, h, h->async);
            h->async->compute_worker = std::thread(compute_loop, h, h->async);
            h->async->completer = std::thread(completer_loop, h, h->async);
        }
        P = h->async;
    }
    auto t = std::make_unique<acl_ticket>();
    t->seq = g_trace.seq.fetch_add(1, std::memory_order_relaxed);
    g_trace.mark(t->seq, "submit");
    t->items = items;
    t->n = n;
    t->perm = perm_out;
    t->err = err_out;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        // chip-filling batches that have to be COPIED go through the one-worker pipeline (look-ahead H2D, kernels chained on the device); everything
        // else -- also every batch the kernel answers across PCIe itself, whatever its size -- is a whole blocking call on any worker: such
        // calls need no turn-taking (profiles/r03_hostmapped_batches.txt)
        if (n >= kComputeTokenItems && n <= h->max_sub_batch && h->shard.world == 1 && !hostmap_takes(h, n)) P->compute.push_back(t.get());
        else P->queue.push_back(t.get());
    }
    P->cv.notify_all();
    *ticket_out = t.release();
    return ACL_OK;
}

int acl_ticket_wait(acl_engine_t *h, acl_ticket_t *tp) {
    if (!tp) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_ticket_wait: NULL ticket");
    std::unique_ptr<acl_ticket> t(tp);
    int rc;
    std::string msg;
    g_trace.mark(t->seq, "wait_begin");
    {
        std::unique_lock<std::mutex> lk(t->mu);
        t->cv.wait(lk, [&] { return t->done; });
        g_trace.mark(t->seq, "wait_returns");
        rc = t->rc;
        msg = t->msg;
    }
    return rc ? fail(rc, msg) : ACL_OK;
}

int acl_host_alloc(acl_engine_t *h, size_t bytes, void **out) {
    if (!out) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_host_alloc: out is NULL");
    *out = nullptr;
    if (h->store_only) return fail(ACL_ERR_UNAVAILABLE, "engine was opened store-only (no GPU)");
    HIP_TRY(hipSetDevice(h->dev0().device));
    void *p = nullptr;
    // (portable + mapped: the kernels of EVERY replica read items from, and write answers to, these buffers across PCIe)
    HIP_TRY(hipHostMalloc(&p, std::max<size_t>(bytes, 64), h->devs.size() > 1 ? (hipHostMallocPortable | hipHostMallocMapped) : hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(h->pinned_mu);
    h->pinned.emplace_back((uintptr_t)p, std::max<size_t>(bytes, 64));
    *out = p;
    return ACL_OK;
}

int acl_host_free(acl_engine_t *h, void *p) {
    if (!p) return ACL_OK;
    {
        std::lock_guard<std::mutex> lk(h->pinned_mu);
        auto it = std::find_if(h->pinned.begin(), h->pinned.end(), [&](const std::pair<uintptr_t, size_t> &r) { return r.first == (uintptr_t)p; });
        if (it == h->pinned.end()) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_host_free: not a buffer from acl_host_alloc");
        h->pinned.erase(it);
    }
    HIP_TRY(hipHostFree(p));
    return ACL_OK;
}

}  // extern "C"
