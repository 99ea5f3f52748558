// engine_internal.hpp -- shared by the translation units behind the C ABI (engine.cpp: core + Check/Filter,
// engine_shard.cpp: acl_shard_*, engine_callers.cpp: keep mask / bitmap test / watch / micro-batcher,
// engine_async.cpp: submit/wait + pinned host buffers).
//
// Threading model (the seam is called from arbitrary goroutines: reference pkg/authz/check.go:77-93,
// responsefilterer.go:165, watch.go:50):
//   state_mu  writer-preferring RW lock over the relationship store AND the snapshot (host copy + device arrays).
//             Evaluations (Check / LookupResources) hold it SHARED for the whole call; writes and snapshot
//             maintenance (patch / rebuild / upload) hold it EXCLUSIVE -- so a patch never overwrites rows a kernel reads.
//   names_mu  schema + object-name tables.  string -> id lookups take it shared (and nothing else: callers of the string
//             entry points intern in parallel, outside any evaluation lock); whatever adds names takes it exclusive
//             while holding state_mu (shared or exclusive).  Lock order: state_mu, then names_mu.
//   PassCtx   everything ONE in-flight evaluation needs on the device: its own stream, frontier buffers, status block,
//             per-batch scratch, pinned staging.  A pool of them lets concurrent callers overlap: while one call waits
//             for its level burst, another's H2D copy, kernels or D2H copy run (verdict r1: one engine-wide mutex
//             serialised everything).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aclgpu.h"
#include "kernels.hpp"
#include "plan.hpp"
#include "store.hpp"

using namespace acl;

struct DevState;  // one replica of the HBM snapshot (below)

namespace aclint {

extern thread_local std::string g_last_error;  // engine.cpp
extern thread_local int g_last_detail;          // what KIND of failure the last fail() was, for callers inside the library that react to one (0: nothing special)
constexpr int kDetailBitmapTooSmall = 1;        // a lookup's caller-sized row no longer covers the type's ids: acl_lookup_resources_alloc sizes again and retries

inline int fail(int code, const std::string &msg) {
    g_last_error = msg;
    g_last_detail = 0;
    return code;
}
inline int fail_detail(int code, int detail, const std::string &msg) {
    g_last_error = msg;
    g_last_detail = detail;
    return code;
}
inline int fail(const Status &s) { return fail(s.code, s.msg); }

#define HIP_TRY(expr)                                                                                           \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return fail(ACL_ERR_INTERNAL, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Reader/writer lock with writer preference (a WriteRelationships must not starve behind a stream of Checks) whose SHARED side is a plain
// count: whoever finishes an evaluation may release it, not only the thread that took it.  Pipelined tickets take the lock on the staging
// thread and give it back on the thread that completes the batch (engine_async.cpp); with a pthread rwlock that hand-over was undefined
// behaviour that glibc happened to tolerate.  One mutex acquisition per call -- calls are device passes, not single loads.
class RwLock {
  public:
    RwLock() = default;
    RwLock(const RwLock &) = delete;
    RwLock &operator=(const RwLock &) = delete;
    void lock() {
        std::unique_lock<std::mutex> lk(m_);
        waiting_writers_++;
        wcv_.wait(lk, [&] { return !writer_ && readers_ == 0; });
        waiting_writers_--;
        writer_ = true;
    }
    void unlock() {
        {
            std::lock_guard<std::mutex> lk(m_);
            writer_ = false;
        }
        wcv_.notify_one();
        rcv_.notify_all();
    }
    void lock_shared() {
        std::unique_lock<std::mutex> lk(m_);
        rcv_.wait(lk, [&] { return !writer_ && waiting_writers_ == 0; });
        readers_++;
    }
    bool try_lock_shared() {  // (fails while a writer holds the lock or waits for it)
        std::lock_guard<std::mutex> lk(m_);
        if (writer_ || waiting_writers_) return false;
        readers_++;
        return true;
    }
    void unlock_shared() {
        bool last;
        {
            std::lock_guard<std::mutex> lk(m_);
            last = --readers_ == 0;
        }
        if (last) wcv_.notify_one();
    }

  private:
    std::mutex m_;
    std::condition_variable rcv_, wcv_;
    uint32_t readers_ = 0, waiting_writers_ = 0;
    bool writer_ = false;
};

// Bits set in n 32-bit words.  LookupResources counts the ids of every result bitmap on the host (64 lookups x 12 KB per C3 step): the
// portable __builtin_popcount loop cost 190 us of a 500 us step; the popcnt instruction over 64-bit words, 25 us.
__attribute__((target("popcnt"))) inline uint64_t popcount_words_hw(const uint32_t *d, size_t n) {
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    size_t w = 0;
    for (; w + 8 <= n; w += 8) {
        uint64_t v[4];
        std::memcpy(v, d + w, 32);
        c0 += (uint64_t)__builtin_popcountll(v[0]);
        c1 += (uint64_t)__builtin_popcountll(v[1]);
        c2 += (uint64_t)__builtin_popcountll(v[2]);
        c3 += (uint64_t)__builtin_popcountll(v[3]);
    }
    for (; w < n; w++) c0 += (uint64_t)__builtin_popcount(d[w]);
    return c0 + c1 + c2 + c3;
}
inline uint64_t popcount_words_portable(const uint32_t *d, size_t n) {
    uint64_t c = 0;
    for (size_t w = 0; w < n; w++) c += (uint64_t)__builtin_popcount(d[w]);
    return c;
}
inline uint64_t popcount_words(const uint32_t *d, size_t n) {
    static const bool hw = __builtin_cpu_supports("popcnt") != 0;
    return hw ? popcount_words_hw(d, n) : popcount_words_portable(d, n);
}

template <typename T>
struct DevArray {
    T *p = nullptr;
    size_t n = 0;
    DevArray() = default;
    DevArray(const DevArray &) = delete;
    DevArray &operator=(const DevArray &) = delete;
    ~DevArray() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void swap(DevArray &o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
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

// pinned host staging (grow-only)
struct PinnedBuf {
    void *p = nullptr;
    void *dp = nullptr;  // the device's pointer to the same memory (hipHostGetDevicePointer, asked once per allocation: the small-call path asked four times per call)
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    hipError_t ensure(size_t bytes) {
        if (bytes <= n && p) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
        dp = nullptr;
        hipError_t e = hipHostMalloc(&p, std::max<size_t>(bytes, 64), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, p, 0);
        if (e == hipSuccess) n = bytes;
        return e;
    }
};

constexpr size_t kComputeTokenItems = 32768;  // host-buffer batches at least this large run their kernels one batch at a time

// per-call cancellation / deadline (reference: LookupResources runs on the HTTP request's ctx and is abandoned when it is
// cancelled, responsefilterer.go:165-170; the prefilter join times out after 10 s, responsefilterer.go:44,196-204)
struct CallOpts {
    const volatile int32_t *cancel = nullptr;  // *cancel != 0: abandon the call
    int64_t deadline_ns = 0;                   // CLOCK_MONOTONIC nanoseconds; 0 = none
};
int64_t mono_ns();
// ACL_OK, ACL_ERR_CANCELLED or ACL_ERR_DEADLINE_EXCEEDED
int check_opts(const CallOpts &o);

// Device state of ONE in-flight evaluation.
struct PassCtx {
    int index = 0;           // among the contexts of ITS device
    DevState *dev = nullptr;  // the replica (device) this context's stream and buffers live on
    hipStream_t stream = nullptr;
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};  // further streams of a host batch cut into concurrent slices (engine.cpp check_pass_local_host), made on first use
    // frontier
    DevArray<uint4> d_fbuf[2];
    DevArray<uint32_t> d_fcounts[2], d_status;  // status = nchunks[kLevelSlots] | any[kLevelSlots] | overflow | export counters
    uint64_t frontier_entries = 0;
    uint32_t max_chunks = 0;
    uint32_t *h_status = nullptr;  // pinned
    DevArray<uint32_t> d_ccount;   // [2] sharded graph, schemas with `&` / `-`: {leaf cells handed out, nodes appended} of this shard (the unsharded level loop counts in the status block)
    DevArray<uint4> d_nodes_all;   // ... the shards' node lists, all-gathered behind the walk
    uint32_t shard_pool_shift = 0; // ... log2 of how far the node / cell pools have been grown beyond their first size (kOverflowPools)
    DevArray<uint32_t> d_done;     // [1] arrival counter of a small host batch's blocks (zero between launches: the last block re-arms it; kernels.hip done_flag)
    uint32_t done_seq = 0;         // the value the next such launch stores into its pinned completion word
    // batch scratch
    DevArray<uint8_t> d_has, d_err, d_perm;
    DevArray<int32_t> d_errout;
    DevArray<uint4> d_items;
    DevArray<uint32_t> d_sids, d_visited, d_rows;
    // result rows beyond the block's LDS (kernels.hpp RevBigRows): the lookups' deferred terminal rows, their counts / levels, the id-count accumulators
    DevArray<uint8_t> d_big_bytes;  // the lookups' byte maps, zero between launches (bytes_zeroed: the leading bytes known to be)
    size_t big_bytes_zeroed = 0;
    DevArray<uint64_t> d_big_tasks, d_big_counts;
    DevArray<uint32_t> d_big_meta;  // [2 m]: task counts, levels
    size_t big_counts_zeroed = 0;   // leading accumulators known to be zero (k_rev_rows re-arms what it used)
    bool direct_tripped = false;   // this context's last walk gave up on the direct task lists (kOverflowDirect): its redo is not a frontier overflow (walk_outcome)
    size_t visited_zero_words = 0;  // leading words of d_visited known to be all zero: k_rev_local takes `visited` zeroed and hands it back zeroed (kernels.hip)
    DevArray<uint4> d_nodes;     // schemas with `&` / `-`: the CombineNode records of a pass (plan.hpp)
    DevArray<uint64_t> d_dedup;  // duplicate-merging passes only (check_pass): open-addressing table over one level's entries
    PinnedBuf h_in, h_out;  // staging for pageable caller buffers
    // native sharded loop (engine_shard_native.cpp): exchange blocks [header | xcap entries], per-level control records
    DevArray<uint4> d_xsend, d_xrecv, d_xhsend, d_xhrecv;  // entries (cap per block) and their 16-byte headers
    DevArray<uint32_t> d_xctrl;
    PinnedBuf h_xctrl;
    uint32_t xcap = 0;
    std::vector<uint8_t> xplan_fwd, xplan_rev;  // [iteration] the previous batch exported something there: exchange entries (empty: always)
    uint32_t rev_levels_hint = 4;
    uint32_t levels_hint = 6;
    CallOpts opts;  // of the call that holds this context
    // measurement (merged into the engine's stats when the context is released)
    acl_stats_t stats{};
    bool timing = false;
    std::vector<hipEvent_t> ev;  // pairs
    size_t ev_used = 0;
    std::vector<int> ev_kind;  // per pair: 0 other, 1 expand

    ~PassCtx();
};

}  // namespace aclint
using namespace aclint;

struct AsyncPool;
namespace aclint {
struct InternPool;  // engine.cpp
}  // engine_async.cpp

// Background snapshot compaction (engine.cpp).  Patching writes into the HBM snapshot leaves garbage behind (relocated
// rows) and eats the tables' headroom; once either passes a threshold the next snapshot is built from a copy-on-write view
// of the store on a worker thread and uploaded to fresh device arrays on the worker's own stream, while reads keep
// patching and using the old one.  The next reader adopts it under the lock: catch-up patch from the view's revision,
// pointer swap.  (Round 1 rebuilt synchronously: the first read after the threshold paid 100-140 ms on the 10 M graph.)
struct Compaction {
    std::thread worker;
    std::atomic<int> state{0};  // 0 idle, 1 building, 2 ready, 3 failed
    Snapshot snap;
    ShardSpec shard;
    bool with_reverse = false;
    int64_t now = 0;
    struct PerDevice {  // one fresh set of snapshot arrays per replica, uploaded on a stream of that device
        int device = 0;
        hipStream_t stream = nullptr;
        DevArray<uint32_t> d_meta, d_edges, d_buckets, d_tsb, d_tnm, d_rmeta, d_redges, d_sbb, d_snobj;
        DevArray<FwdOp> d_ops;
        DevArray<SlotProg> d_progs;
        DevArray<uint32_t> d_bexpr;
        DevArray<RevOp> d_rops;
        DevArray<RevProg> d_rprogs, d_rseeds;
        DevArray<uint64_t> d_rdest;
    };
    std::vector<std::unique_ptr<PerDevice>> per;  // [replica]
    std::string error;
};

// One REPLICA of the HBM snapshot: a device, its copy of the snapshot arrays, the evaluation contexts (streams) on it.  An engine opened on
// several devices (acl_config_t.devices) is ONE relationship store, ONE set of name tables and ONE host snapshot in front of N of these: the
// reference holds one PermissionsClient per process (pkg/proxy/options.go:371-377) and the dual-write worker shares it (server.go:136-153).
// Every snapshot update (patch / rebuild / adoption of a background build) reaches EVERY replica under state_mu exclusive before the lock is
// released, so whichever replica an evaluation lands on answers for the store as it is: read-your-writes holds per process, not per device.
struct DevState {
    int device = 0;  // HIP ordinal
    int index = 0;   // among the engine's replicas
    bool dev_valid = false, rev_uploaded = false;  // the device arrays hold exactly the host snapshot / its reverse rows
    hipStream_t up_stream = nullptr;               // snapshot uploads (always under state_mu exclusive)
    int grid_blocks = 2048;
    int local_blocks = 1024;       // resident blocks of the single-launch kernel (4 waves per block)
    int local_blocks_wide = 512;   // ... of its wide (12-wave) instantiation
    // forward graph
    DevArray<uint32_t> d_meta, d_edges, d_buckets, d_tsb, d_tnm;
    DevArray<FwdOp> d_ops;
    DevArray<SlotProg> d_progs;
    DevArray<uint32_t> d_bexpr;  // boolean programs of the combine slots (schemas with `&` / `-`)
    // reverse graph
    DevArray<uint32_t> d_rmeta, d_redges, d_sbb, d_snobj;
    DevArray<RevOp> d_rops;
    DevArray<RevProg> d_rprogs, d_rseeds;
    DevArray<uint64_t> d_rdest;
    // evaluation contexts of this device (the pool's lock and condition variable are the engine's)
    std::vector<std::unique_ptr<PassCtx>> ctxs;  // created lazily up to max_ctx
    std::vector<PassCtx *> free_ctxs;
    uint32_t in_use = 0;  // contexts handed out (Eval::begin spreads calls over the replicas by it)
    uint64_t calls = 0;   // evaluations this replica has been handed since open (acl_replica_calls; under pool_mu)
    std::mutex compute_mu;  // turn-taking of chip-filling batches on the level loop (check_ids_host's fallback, the submit/wait pipeline)
};

struct acl_engine {
    RwLock state_mu;             // store + snapshot (see the header comment)
    std::shared_mutex names_mu;  // schema + object-name tables
    Store store;
    Snapshot snap;
    ShardSpec shard;  // world > 1: this engine holds one shard of the graph and only the acl_shard_* entry points evaluate
    // snap_valid: the HOST snapshot matches `snap.revision`; dev_valid: the device arrays hold exactly the host snapshot.
    // Both are cleared before the snapshot is touched and set again only after every upload succeeded (advice r1: a failed
    // upload used to leave a "valid" snapshot behind).
    bool snap_valid = false;
    std::vector<std::unique_ptr<DevState>> devs;  // the replicas: one per entry of acl_config_t.devices (default: one, on acl_config_t.device)
    DevState &dev0() { return *devs[0]; }          // (the sharded entry points and the test hooks run on the first replica)
    bool all_dev_valid() const {
        for (const auto &d : devs)
            if (!d->dev_valid) return false;
        return !devs.empty();
    }
    bool all_rev_uploaded() const {
        for (const auto &d : devs)
            if (!d->rev_uploaded) return false;
        return !devs.empty();
    }
    void set_dev_valid(bool v) {
        for (auto &d : devs) d->dev_valid = v;
    }
    void set_rev_uploaded(bool v) {
        for (auto &d : devs) d->rev_uploaded = v;
    }
    std::atomic<uint64_t> keep_route_calls{0};  // PostFilter calls answered by ONE reverse walk + bit tests (engine.cpp keep_by_reverse_walk)
    // ... and what the last such call for a (type, permission, subject) found: 58 bits of the key's hash | 1 + the bit width of the allowed count (0: never seen).
    // A hint only -- it decides whether the host's pass resolves names while the device still walks -- and direct-mapped: a collision costs a wrong guess.
    std::atomic<uint64_t> keep_seen[256] = {};
    std::atomic<uint32_t> keep_route_skips{0};  // short lists for far-reaching subjects that went forward instead (one in sixteen still walks)
    // The pair form of that route on a RECURSIVE permission (Snapshot::slot_deep: the schema alone cannot rule a depth error out) needs to know that THE DATA rules it
    // out: no object of the type whose Check runs into the dispatch-depth limit.  That is a property of the snapshot, not of the subject (engine.cpp
    // no_object_is_deep): one forward sweep over the type's objects for a subject nobody is, remembered per (type, permission, subject type) and snapshot epoch.
    uint64_t snap_epoch = 0;  // counts the snapshot's changes (ensure_snapshot, under state_mu exclusive; read under state_mu shared)
    struct DeepKnown {
        int rt = -1, pm = -1, st = -1;
        bool swept = false, none = false;  // none: no object of rt whose Check of pm for a subject of type st ends at the depth limit ...
        uint64_t epoch = 0, adds = 0;      // ... on the snapshot of this epoch, the store's path_adds() then.  `none` stays true while no write ADDS a path
                                           // (Store::path_adds: plain-subject relationships and removals cannot make a Check deeper); "some object is
                                           // deep" is only known for its own epoch (a DELETE may have cut the cycle)
        uint64_t wanted = 0;               // the key (path_adds, or the epoch after a sweep that found deep objects) at which a call last asked and went forward
        bool wanted_by_epoch = false;
        std::shared_ptr<const std::vector<uint32_t>> bits;  // !none: the deep objects of rt (a bit per id), for `epoch`
    };
    std::mutex deep_mu;
    std::vector<DeepKnown> deep_known;  // (a handful: one per list rule's template)
    std::atomic<uint64_t> depth_sweeps{0};
    bool per_item_validation = false;  // ACL_FLAG_PER_ITEM_VALIDATION: ill-formed items of a bulk Check fail their own pair, not the call
    bool lenient_lookup = false;       // ACL_FLAG_LENIENT_LOOKUP: a LookupResources candidate whose forward Check errs is dropped instead of failing the call
    bool store_only = false;  // ACL_FLAG_STORE_ONLY: relationship store without a device (reads that need the GPU fail)
    // the single-launch walk met rows too long for its direct task lists on this snapshot (kOverflowDirect): later walks build their lists the general
    // way (reset when a snapshot is rebuilt); (the batch that found out is redone without a back-off: PassCtx::direct_tripped, walk_outcome)
    std::atomic<bool> walk_no_direct{false};
    std::atomic<int> local_skip{0}, local_fail_streak{0};  // large passes the walk sits out after it overflowed (check_pass)
    bool raw_intern = false;       // test knob (ACL_RAW_INTERN): acl_intern skips the API's object-id pattern
    uint32_t local_cap_limit = 0;  // test knob (ACL_LOCAL_CAP): private frontier entries per block, at most
    uint64_t compaction_slack = 65536;  // words of garbage a snapshot may hold on top of an eighth of its rows before a background build starts
    uint32_t spin_max = 64;  // (every block pays a system-scope release of its answers: worth it up to 64 blocks -- 1 item 21.3 -> 17.6 us, 64 items 23.0 -> 20.6 us; at 256 items it already loses, 23 -> 27 us, profiles/r05_spin_wait_ab.txt) host batches up to this size wait for their kernel by spinning on a pinned word its last block stores, not in hipStreamSynchronize (ACL_SPIN_MAX; 0 = off)
    uint32_t hostmap_max = 0xFFFFFFFFu;  // host batches up to this size: the kernel reads the items from, and writes the answers to, pinned host memory (no copies; ACL_HOSTMAP_MAX, A/B knob)
    unsigned intern_threads = 32; // host threads (the caller included) of bulk string interning, at most
    uint32_t local_wide_min = 65536;  // batches from this size on run the wide (12-wave) instantiation (a unit pools more requests: shorter tail)
    uint32_t host_skew_pct = 8;  // a lone caller's host-mapped launch: first unit this many percent larger than the mean, last one as much smaller (ACL_HOST_SKEW_PCT;
                                 // worth 1-3 % of such a call -- the items do NOT arrive in block order, or 16 % would have hidden half the transfer: profiles/r04_host_skew.txt)
    uint32_t host_split = 2;   // streams the slices of one large host batch are spread over (ACL_HOST_SPLIT, 1 = off; engine.cpp check_pass_local_host)
    uint32_t local_upw = 1;    // single-launch pass over a large batch: work units per resident wave.  1 = every wave one unit of n / waves requests (no
                               // second round of per-level latency chains); 2 balances C4's uneven requests 3 % better but costs C2 a whole second round
    uint32_t local_static_pct = 100, local_dyn_unit = 32;  // chip-filling single-launch passes: share of the batch in static (one per block) units; hand-out unit size
    uint64_t cfg_frontier_entries = 0;
    // evaluation contexts: per replica (DevState), one lock and one condition variable for all
    std::mutex pool_mu;
    std::condition_variable pool_cv;
    uint32_t max_ctx = 4;  // per replica
    size_t next_dev = 0;   // where Eval::begin starts looking (under pool_mu)
    std::unique_ptr<PassCtx> shard_ctx;  // the acl_shard_* protocol keeps state across calls: its own context, never pooled
    void *rccl_comm = nullptr;           // ncclComm_t of acl_shard_rccl_init
    std::unique_ptr<Compaction> compaction;  // touched only under state_mu exclusive (the worker owns its innards while state == 1)
    bool compaction_enabled = true;
    std::mutex shard_mu;
    // change-feed waiters (acl_watch_wait): every committed write bumps feed_gen under feed_mu and wakes them
    std::mutex feed_mu;
    std::condition_variable feed_cv;
    uint64_t feed_gen = 0;
    void feed_wake() {
        {
            std::lock_guard<std::mutex> lk(feed_mu);
            feed_gen++;
        }
        feed_cv.notify_all();
    }
    uint32_t max_sub_batch = 1u << 20;
    uint32_t local_max_items = 1u << 20;  // batches up to this size take the single-launch path (k_check_local) first; 0 = never.  Measured on C4
                                          // (profiles/r02_walk_vs_levels.txt): faster than the level loop at every batch size, 1.5x at 262 144 items
    bool rev_rows_device = false;  // k_rev_local's result rows through a device buffer + one DMA copy instead of kernel writes to host memory (A/B)
    bool shard_a2a = true;  // native sharded Check: per-destination blocks through the communicator's all_to_all when it has one (ACL_SHARD_A2A=0: all-gather)
    bool rev_sink_on = true;   // a lookup's result slot that nothing above leads back into is marked, not expanded (Snapshot::rev_sink; ACL_REV_SINK=0: A/B and tests)
    bool rev_lds_rows = true;  // k_rev_local keeps the result slot's rows in LDS when they fit (ACL_REV_LDS_ROWS=0: always in HBM; A/B and tests)
    uint32_t rev_defer_min = 0;  // 0 = the kernels' default (4096 children per round); ACL_REV_DEFER_MIN: test knob
    bool rev_big_rows = true;   // result rows beyond the LDS: deferred terminal rows + chip-wide row copy (kernels.hip RevDefer); ACL_REV_BIG_ROWS=0 switches it off (A/B)
    bool rev_local = true;  // LookupResources: the single-launch reverse walk (k_rev_local) first; ACL_REV_LOCAL=0 = always the level loop (A/B)
    uint32_t lk_target = 0;  // sharded lookup in flight: target slot, number of requests
    size_t lk_n = 0;
    // micro-batching front-end (engine_callers.cpp)
    struct Batcher;
    Batcher *batcher = nullptr;  // lives from acl_open to acl_close (batcher_create / batcher_destroy); start / stop only toggle its threads
    std::mutex batcher_mu;       // start / stop
    // async submit / wait (engine_async.cpp)
    AsyncPool *async = nullptr;  // created by the first submit, destroyed by async_shutdown
    aclint::InternPool *intern_pool = nullptr;  // host threads of bulk string interning (engine.cpp), created by the first large string batch
    std::mutex intern_pool_mu;
    std::mutex async_mu;
    // pinned buffers handed out by acl_host_alloc: [base, base + bytes)
    std::mutex pinned_mu;
    std::vector<std::pair<uintptr_t, size_t>> pinned;
    // measurement
    std::mutex stats_mu;
    acl_stats_t stats{};
    std::atomic<bool> timing{false};


    DevGraph dev_graph(const DevState &d) const {
        DevGraph g{d.d_meta.p, d.d_edges.p, d.d_buckets.p, d.d_ops.p, d.d_progs.p, d.d_tsb.p, d.d_tnm.p, snap.nslots, snap.ntypes, (uint32_t)snap.ops.size()};
        g.walk_flags = walk_no_direct.load(std::memory_order_relaxed) ? kWalkNoDirect : 0u;
        return g;
    }
    DevGraph dev_graph(const PassCtx *c) const { return dev_graph(*c->dev); }
    DevReverse dev_reverse(const PassCtx *c, uint32_t vwords) const {
        const DevState &d = *c->dev;
        return DevReverse{d.d_rmeta.p, d.d_redges.p, d.d_rops.p, d.d_rprogs.p, d.d_rseeds.p, d.d_rdest.p, d.d_sbb.p, d.d_snobj.p, c->d_visited.p, vwords,
                          (uint32_t)snap.rprogs.size(), (uint32_t)snap.rops.size()};
    }
    DevFrontier dev_frontier(const PassCtx &c) const {
        DevFrontier f;
        f.buf[0] = c.d_fbuf[0].p;
        f.buf[1] = c.d_fbuf[1].p;
        f.counts[0] = c.d_fcounts[0].p;
        f.counts[1] = c.d_fcounts[1].p;
        f.nchunks = c.d_status.p;
        f.any = c.d_status.p + kLevelSlots;
        f.overflow = c.d_status.p + 2 * kLevelSlots;  // [+1]: the level's export counter (sharded graph)
        f.nwaves = (uint32_t)c.dev->grid_blocks * kWavesPerBlock;
        f.max_chunks = c.max_chunks;
        return f;
    }
    bool is_pinned(const void *p, size_t bytes);
};

namespace aclint {

int alloc_frontier(acl_engine *h, PassCtx *c, uint64_t entries);
void ev_begin(PassCtx *c, int kind);
void ev_end(PassCtx *c);
void ev_collect(PassCtx *c);  // stream must be synchronized
// snapshot maintenance; caller holds state_mu EXCLUSIVE
int ensure_snapshot(acl_engine *h);
int ensure_reverse(acl_engine *h);
// true when the device snapshot answers for the store as it is now (caller holds state_mu at least shared)
bool snapshot_current(acl_engine *h, bool need_reverse);
void compaction_join(acl_engine *h);  // acl_close / schema reload: waits for a build in flight and drops its result

// RAII for one evaluating call: state_mu shared (snapshot brought up to date first) + a PassCtx from the pool.
struct Eval {
    acl_engine *h = nullptr;
    PassCtx *c = nullptr;
    bool locked = false;
    Eval() = default;
    Eval(const Eval &) = delete;
    Eval &operator=(const Eval &) = delete;
    ~Eval() { end(); }
    // rev_key_slot >= 0: the lookup's subject is `type#relation` of that slot -- the reverse rows must cover its id space
    // on_device >= 0: only a replica on that HIP device will do (calls that are handed device pointers)
    int begin(acl_engine *h_, bool need_reverse, const CallOpts &opts = CallOpts(), int rev_key_slot = -1, int on_device = -1);
    void end();
};

// one call of the acl_shard_* step protocol (engine_shard.cpp): shard_mu + state_mu shared + the shard's own context
struct ShardCall {
    acl_engine *h = nullptr;
    PassCtx *c = nullptr;
    bool locked = false, have_mu = false;
    ShardCall() = default;
    ShardCall(const ShardCall &) = delete;
    ShardCall &operator=(const ShardCall &) = delete;
    ~ShardCall();
    int begin(acl_engine *h_, bool fresh, bool need_reverse, bool combine_ok = false /* the caller evaluates schemas with `&` / `-` (the native Check loop) */);
};
DevShard dev_shard(acl_engine *h, PassCtx *c, void *d_export, size_t cap);
int new_ctx(acl_engine *h, DevState *d, std::unique_ptr<PassCtx> *out, int index);  // (leaves the calling thread on d's device)
void merge_stats(acl_engine *h, PassCtx *c);

// schemas with `&` / `-`: sizes the pass's node list and result cells (has[] / err[] behind the n requests' own) and points g at them.
// blocks > 0: the single-launch walk (per-block regions for units of rpw requests); 0: the level loop (one pool, counters in the status block)
int combine_prepare(acl_engine *h, PassCtx *c, DevGraph *g, uint32_t n, uint32_t blocks, uint32_t rpw);
int check_pass(acl_engine *h, PassCtx *c, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout, bool try_local = true);
// every level in ONE launch, wave-private frontiers; an internal negative code when a wave's private frontier overflowed (engine.cpp)
int check_pass_local(acl_engine *h, PassCtx *c, const DevGraph &g, const uint4 *d_items, uint32_t n, uint8_t *d_perm, int32_t *d_errout);
int not_sharded(acl_engine *h);
int device_of(acl_engine *h, const void *p);  // HIP ordinal a device pointer lives on; -1 = any replica will do
int check_device(acl_engine *h, PassCtx *c, const uint4 *d_items, size_t n, uint8_t *d_perm, int32_t *d_errout, bool try_local = true);  // try_local: the single-launch walk first
// host items -> answers in host buffers through context c (pinned staging unless the caller's buffers are pinned)
int check_ids_host(acl_engine *h, PassCtx *c, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, bool items_on_device = false);  // items_on_device: c->d_items already holds them (items is not read)
int lookup_candidate_error(acl_engine *h, int32_t code, uint32_t id, uint32_t sid);  // fails a LookupResources call with a candidate's Check error (lookups.go:75-83)
int lookup_batch(acl_engine *h, PassCtx *c, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n, uint32_t *bitmaps, size_t words,
                 uint64_t *counts);
bool empty(const char *s);
FilterText to_filter(const acl_filter_t *f);
// strings -> interned item; returns 0 or the per-item error the pair carries (check.go:55).  Caller holds names_mu shared.
int32_t intern_check_item(acl_engine_t *h, const acl_check_item_t &it, acl_item_t *out);
void intern_pool_destroy(acl_engine_t *h);
void host_parallel(acl_engine_t *h, size_t total, size_t piece, const std::function<void(size_t, size_t)> &fn);  // engine.cpp: pieces of [0, total) on the interning pool's threads + the caller
unsigned host_threads(acl_engine_t *h);
bool hostmap_takes(acl_engine *h, size_t n);  // engine.cpp: a host batch of n items is answered by the kernel across PCIe (no copies)
int resolve_lookup(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, int *rt_out, int *pm_out,
                   int *st_out, int *sr_out, uint32_t *sub_out);
int check_bulk_keep_cstr_call(acl_engine_t *h, const acl_check_item_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out);  // engine.cpp: acl_check_bulk_keep
int lookup_batch_call(acl_engine_t *h, int rtype, int perm, int stype, int srel, const uint32_t *sids, size_t n, uint32_t *bitmaps, size_t words,
                      uint64_t *counts, const CallOpts &opts);
int lookup_one_routed(acl_engine_t *h, int rt, int pm, int st, int sr, uint32_t sub, uint32_t *bitmap_out, size_t words, uint64_t *count_out,
                      const CallOpts &opts);
void async_shutdown(acl_engine_t *h);
void batcher_create(acl_engine_t *h);
void batcher_destroy(acl_engine_t *h);  // after acl_batcher_stop
int lookup_opts_call(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint32_t *bitmap_out,
                     size_t bitmap_words, uint64_t *count_out, const CallOpts &opts);

// Runs iterations 1.. of a level loop until the frontier is empty.  `launch(iter)` enqueues
// one expansion.  Returns ACL_OK, or ACL_ERR_RESOURCE_EXHAUSTED when the frontier overflowed.
// `tail()` is enqueued after every burst, BEFORE the host learns whether the burst reached the last level: when it did
// (the common case -- the burst is sized by the previous batch's depth) the batch's epilogue has already run by the time
// the status read-back completes, instead of costing another launch + sync round trip; when it did not, the epilogue
// simply runs again after the next burst (it only reads the final has/err).
// Cancellation / deadline (PassCtx::opts) is honoured between bursts.
template <typename F, typename T>
int level_loop(acl_engine *h, PassCtx *c, uint32_t max_iter, F launch, uint32_t *levels_out, T tail) {
    const bool watched = c->opts.cancel || c->opts.deadline_ns;
    uint32_t next = 1, burst = std::max<uint32_t>(c->levels_hint, 2);
    if (watched) burst = std::min<uint32_t>(burst, 2);  // a watched call learns about its cancellation every two levels
    for (;;) {
        if (watched) {
            int rc = check_opts(c->opts);
            if (rc) {
                (void)hipStreamSynchronize(c->stream);
                return rc;
            }
        }
        uint32_t last = std::min(max_iter, next + burst - 1);
        for (uint32_t it = next; it <= last; it++) {
            ev_begin(c, 1);
            launch(it);
            ev_end(c);
            c->stats.expand_launches++;
        }
        tail();
        HIP_TRY(hipMemcpyAsync(c->h_status, c->d_status.p, kStatusWords * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        ev_collect(c);
        if (c->h_status[2 * kLevelSlots] == 2) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "a relationship row exceeds the per-task enumeration limit");
        if (c->h_status[2 * kLevelSlots] == 3) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "the batch visits more states with intersection / exclusion rewrites than one pass holds: lower max_sub_batch");
        if (c->h_status[2 * kLevelSlots]) return ACL_ERR_RESOURCE_EXHAUSTED;
        uint32_t done_at = 0;
        for (uint32_t it = next; it <= last; it++)
            if (c->h_status[kLevelSlots + it] == 0) { done_at = it; break; }  // any[it]: iteration `it` produced nothing
        if (done_at || last == max_iter) {
            uint32_t lv = done_at ? done_at : max_iter;
            for (uint32_t it = 0; it < lv; it++) c->stats.frontier_entries += (uint64_t)c->h_status[it] * kChunk;  // dynamic chunks only (lower bound)
            *levels_out = lv;
            return ACL_OK;
        }
        next = last + 1;
        burst = watched ? 2 : 4;
    }
}

}  // namespace aclint
