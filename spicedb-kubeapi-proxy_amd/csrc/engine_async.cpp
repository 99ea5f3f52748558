// engine_async.cpp -- acl_check_bulk_ids_submit / acl_ticket_wait and pinned host buffers.
//
// SURVEY.md 8(d) defines batch throughput on the ABI call that takes HOST ids.  A Go caller keeps several such calls in flight from
// goroutines blocking in acl_check_bulk_ids; submit / wait is the same thing for single-threaded hosts (bench.py --pipeline submit) and for
// callers that want a window of batches in flight: a ticket is a whole blocking call on one of the pool's workers.  Since round 3 the
// kernel reads the items from, and writes the answers to, pinned host memory itself, so such calls need no turn-taking and no copy engine;
// round 4 made that true at every batch size (sub-passes, engine.cpp check_pass_local_host) and retired what this file used to hold for
// batches beyond one launch: look-ahead H2D staging, three "lanes" of contexts, kernels chained on the device through events and a
// completer thread (~250 lines).
#include "engine_internal.hpp"

struct acl_ticket {
    const acl_item_t *items = nullptr;
    size_t n = 0;
    uint8_t *perm = nullptr;
    int32_t *err = nullptr;
    int rc = 0;
    std::string msg;
    bool done = false;
    std::mutex mu;
    std::condition_variable cv;
};

struct AsyncPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<acl_ticket *> queue;
    std::vector<std::thread> workers;
    bool stop = false;
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
        finish(t, acl_check_bulk_ids(h, t->items, t->n, t->perm, t->err));  // (Eval::begin picks the replica and sets this thread's device)
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
    delete P;
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
            const uint32_t nw = std::max<uint32_t>(1, h->max_ctx) * (uint32_t)std::max<size_t>(1, h->devs.size());  // one per evaluation context
            for (uint32_t i = 0; i < nw; i++) h->async->workers.emplace_back(worker_loop, h, h->async);
        }
        P = h->async;
    }
    auto t = std::make_unique<acl_ticket>();
    t->items = items;
    t->n = n;
    t->perm = perm_out;
    t->err = err_out;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->queue.push_back(t.get());
    }
    P->cv.notify_one();
    *ticket_out = t.release();
    return ACL_OK;
}

int acl_ticket_wait(acl_engine_t *h, acl_ticket_t *tp) {
    if (!tp) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_ticket_wait: NULL ticket");
    std::unique_ptr<acl_ticket> t(tp);
    int rc;
    std::string msg;
    {
        std::unique_lock<std::mutex> lk(t->mu);
        t->cv.wait(lk, [&] { return t->done; });
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
