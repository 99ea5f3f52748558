// engine_internal.hpp -- shared by the translation units behind the C ABI (engine.cpp: core + Check/Filter,
// engine_shard.cpp: acl_shard_*, engine_callers.cpp: keep mask / bitmap test / watch / micro-batcher).
#pragma once
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

namespace aclint {


extern thread_local std::string g_last_error;  // engine.cpp

inline int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
inline int fail(const Status &s) { return fail(s.code, s.msg); }

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


}  // namespace aclint
using namespace aclint;

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
        int kind = 0;  // 0: one Check item; 1: one LookupResources request
        acl_item_t item;
        uint8_t perm = 0;
        int32_t err = 0;
        int lk_rtype = 0, lk_perm = 0, lk_stype = 0, lk_srel = -1;  // kind 1
        uint32_t lk_sid = 0;
        uint32_t *lk_bitmap = nullptr;
        size_t lk_words = 0;
        uint64_t lk_count = 0;
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
    uint64_t mb_batches = 0, mb_items = 0, mb_lookup_walks = 0, mb_lookups = 0;
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

namespace aclint {

int alloc_frontier(acl_engine *h, uint64_t entries);
void ev_begin(acl_engine *h, int kind);
void ev_end(acl_engine *h);
void ev_collect(acl_engine *h);  // stream must be synchronized
int ensure_snapshot(acl_engine *h);
int ensure_reverse(acl_engine *h);
int check_pass(acl_engine *h, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout);
int not_sharded(acl_engine *h);
int check_device(acl_engine *h, const uint4 *d_items, size_t n, uint8_t *d_perm, int32_t *d_errout);
bool empty(const char *s);
FilterText to_filter(const acl_filter_t *f);
// strings -> interned item; returns 0 or the per-item error the pair carries (check.go:55)
int32_t intern_check_item(acl_engine_t *h, const acl_check_item_t &it, acl_item_t *out);
int resolve_lookup(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, int *rt_out, int *pm_out,
                   int *st_out, int *sr_out, uint32_t *sub_out);

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

}  // namespace aclint
