// kernels.hip -- gfx950 (CDNA4) frontier-expansion kernels of the batched ACL-check engine.
//
// What they replace: the per-item recursive dispatch SpiceDB runs for CheckBulkPermissions (reached from reference
// pkg/authz/check.go:48 and pkg/authz/postfilter.go:134) and the reverse walk behind LookupResources (pkg/authz/lookups.go:65).
// A pending sub-check is a 16-byte frontier entry (object id, request, slot | level | flags, subject id); a SEGMENT is 64 of them,
// one per lane.  process_segment() advances a segment by one dispatch level:
//   - every lane interprets its state's flattened program (plan.hpp, cached in LDS): probes of the request's subject in hashed rows
//     (4-slot buckets, seeded single-choice: ONE 16 B gather per test -- two-choice rows, two gathers, only where no seed was found;
//     never a probing chain) or sorted rows (binary search), and "enumerate" operations whose child states are produced by a wave-cooperative, load-balanced expansion:
//       - per-lane tasks (row start, degree, the subject's hashed row) are compacted into LDS with wave64 ballot + mbcnt,
//       - a DPP prefix sum over task degrees sizes the work, and every task marks its first work item with a head bit,
//       - lanes take consecutive work items, find their task from the head bits (two v_mbcnt) and load consecutive edges of a row,
//       - each child's own probes are evaluated RIGHT THERE; only children that still have something to enumerate are written,
//         compacted with a second ballot, as consecutive entries.  Leaf states never enter the frontier.
//   - where a segment is uniform enough the interpreter is bypassed: already-probed single-op parents (the deep levels) fetch has[],
//     their row descriptor and the subject's row of the child's probe in ONE trip, two segments at a time; segments of ONE slot build
//     their task list with the prefix sums taken in registers (the direct form, round 5); simple_steps expands the tasks kSimpleWidth
//     children per lane and step; flush_probes handles arrow-target states.
//   Loads are pinned into trips (issue_fence): all loads of a trip are issued before the first of them is waited for.
// Two drivers call it:
//   k_check_local  ONE launch per batch: a block walks its share of the requests through every level inside a block-private
//                  frontier region (LDS cursors, one barrier per level).  The default for every batch size.
//   k_expand       one launch per level over a chunked global frontier (every wave owns one static 16 KiB chunk per level, further
//                  chunks from one atomicAdd per 1024 entries): the sharded graph, and batches that outgrow the private regions.
// The reverse walk (LookupResources): k_rev_local, one block per lookup through every reverse level (result rows in LDS for types up to 1 M objects;
// beyond that k_rev_terminal marks the deferred heavy rows chip-wide and k_rev_rows folds the byte marks into the caller's rows); k_rev_expand is its
// level-loop form (sharded graph, overflows).  k_dedup merges duplicate entries of a level when a frontier explodes.
//
// Bound (profiles/r06_pmc_c4.md, r06_ab_check_local.txt): the CHAIN of dependent trips a wave makes per pair of segments, not HBM and
// (round 6's A/Bs) not instruction issue -- no MFMA anywhere by design.
#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace acl {
namespace {

#ifndef ACL_PERTURB
#define ACL_PERTURB 0  // cost attribution (tools/perturb.sh, profiles/r01_c4_bottleneck_analysis.md): repeat ONE kind of work, results unchanged.
                       // 1 has[] read / entry, 2 row-descriptor gather / entry, 3 hashed probe / child, 5 edge gather / child,
                       // 6 ~60 VALU / child, 7 random bucket gather / child, 8 six LDS reads / child, 9 entry re-read / entry
#endif
#define ACL_KEEP(x) asm volatile("" ::"v"(x))
#ifndef ACL_ANS_LDS
#define ACL_ANS_LDS 1  // with ACL_ENTRY8: the unit's answer bytes (has / err) in LDS (ans_get / ans_set); 0 = in global memory as before (A/B builds)
#endif
#ifndef ACL_ENTRY8
#define ACL_ENTRY8 1  // 8-byte frontier entries in the single-launch walk (put_entry); 0 = the 16-byte form everywhere (A/B builds, tools/build_variant.sh)
#endif
constexpr int kBlock = kWavesPerBlock * 64;
#ifndef ACL_MIN_WAVES_PER_SIMD
#define ACL_MIN_WAVES_PER_SIMD 6  // (k_expand, k_rev_expand; the single-launch kernel has its own bound below)
#endif
constexpr uint32_t kTaskCap = 128;  // LDS task slots per wave
constexpr uint32_t kHeadWords = 32;  // flush_simple maps work items to tasks through head bits: rounds of <= 64 * kHeadWords children
constexpr uint32_t kSelfBit = 0x80000000u;      // task: the child is the same object (start holds its id)
constexpr uint32_t kLeafAuthBit = 0x40000000u;  // task: the row's edges carry authoritative leaf flags
constexpr uint32_t kAllBit = 0x20000000u;       // task (combine schemas): the tupleset of an intersection arrow -- every child gets a result cell of its own (OP_ALL)
constexpr uint32_t kCountMask = 0x1FFFFFFFu;
constexpr uint32_t kMaxRow = 1u << 25;  // rows longer than this cannot be enumerated in one task
constexpr uint32_t kNoSpace = 0xFFFFFFFFu;

// entry meta: slot[0:13) | level[13:19) | probed[19] | subject key[20:32)
constexpr uint32_t kProbedBit = 1u << 19;
__device__ __forceinline__ uint32_t meta_slot(uint32_t m) { return m & 0x1FFFu; }
__device__ __forceinline__ uint32_t meta_level(uint32_t m) { return (m >> 13) & 63u; }
__device__ __forceinline__ uint32_t meta_key(uint32_t m) { return m >> 20; }
__device__ __forceinline__ uint32_t make_meta(uint32_t slot, uint32_t level, uint32_t key) { return slot | (level << 13) | (key << 20); }

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// Everything above this line is issued before anything below it.  Placed between "issue every load of this trip" and "first use":
// left alone, the compiler sinks a load next to its use and waits for it before it issues the next one -- a wave's independent
// gathers then travel one after the other instead of together (seen in the ISA of every multi-load step of this file).
__device__ __forceinline__ void issue_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Wave64 scan / reduction on the DPP lane network: six v_add / v_max with a DPP source operand, no LDS traffic (ds_bpermute)
// and -- what mattered more here -- no per-distance lane-address VGPRs that the compiler hoists and keeps alive across the
// whole kernel.  row_shr:n shifts within a row of 16 lanes (lanes without a source add 0), row_bcast:15 / :31 carry a
// row's / a half's total into the rows above (gfx9 family incl. gfx950).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t) {
    v += dpp_or0<0x111, 0xf>(v);  // row_shr:1
    v += dpp_or0<0x112, 0xf>(v);  // row_shr:2
    v += dpp_or0<0x114, 0xf>(v);  // row_shr:4
    v += dpp_or0<0x118, 0xf>(v);  // row_shr:8  -> inclusive within each row of 16
    v += dpp_or0<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_or0<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {  // same network with max (0 is neutral for unsigned); lane 63 ends up with the maximum
    v = max(v, dpp_or0<0x111, 0xf>(v));
    v = max(v, dpp_or0<0x112, 0xf>(v));
    v = max(v, dpp_or0<0x114, 0xf>(v));
    v = max(v, dpp_or0<0x118, 0xf>(v));
    v = max(v, dpp_or0<0x142, 0xa>(v));
    v = max(v, dpp_or0<0x143, 0xc>(v));
    return wave_last(v);
}

// Phase timing (variant builds only, tools/phases.sh): every wave adds the shader-clock time between two marks to the phase named by
// the second one; loads are waited for AT the mark, so a "wait" phase is the exposed latency of that trip.  ~10 % slower.
#ifndef ACL_PROFILE_PHASES
#define ACL_PROFILE_PHASES 0
#endif
enum { PH_OTHER = 0, PH_ENTRIES, PH_GATHERS, PH_TASKS, PH_PROLOGUE, PH_EDGES, PH_BUCKETS, PH_PUSH, PH_GENERIC, PH_BARRIER, PH_SEED, PH_COUNT };
#if ACL_PROFILE_PHASES
__device__ unsigned long long acl_phase_cycles[16];
#define ACL_MARK(wo, ph)                                                                  \
    do {                                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                       \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();                     \
        if (lane_id() == 0) {                                                             \
            (wo).cold->prof[ph] += now_ - (wo).cold->last;                                \
            (wo).cold->last = now_;                                                       \
        }                                                                                 \
    } while (0)
#else
#define ACL_MARK(wo, ph) do {} while (0)
#endif

// LDS-staged task list of one wave.
#ifndef ACL_TASK_PAD
#define ACL_TASK_PAD 0  // 1: a task record is 20 bytes, not 16 (round 6; VERDICT r5 weak #2: 40 % of the walk's LDS cycles were bank-conflict cycles).  The children of a window
                        // read the records of ~25 consecutive tasks, one dword each: at a 16-byte stride eight records cover the 32 banks and record j meets
                        // j + 8, j + 16, j + 24 in one bank (4-way conflicts on every per-child read); at 20 bytes (5 dwords, coprime with 32) 32 consecutive
                        // records sit in 32 different banks.  MEASURED, same box (profiles/r06_ab_check_local.txt): C4 209.2 us padded against 208.4 us plain, the
                        // replica 280.2 against 279.8 -- the conflicts lengthen LDS accesses that sit in the shadow of the global trips, and the padded
                        // records cost 6 KB of LDS per block.  Off by default; kept as an A/B knob
#endif
struct TaskRec {
    uint32_t x, y, z, w;
#if ACL_TASK_PAD
    uint32_t pad;
#endif
    __device__ __forceinline__ TaskRec &operator=(const uint4 &v) {
        x = v.x; y = v.y; z = v.z; w = v.w;
        return *this;
    }
    __device__ __forceinline__ operator uint4() const { return make_uint4(x, y, z, w); }
};
struct TaskLds {
    TaskRec a[kTaskCap];       // x: first edge (absolute index) -- or the object id for a "self" task; flush_simple turns it into "first edge minus
                               //    first work item" (edge index = x + work item).  y, z: the hashed row the children are probed in (first bucket,
                               //    bucket count; {0, 1} = the reserved empty bucket).  w: the request.  One 16-byte read per child instead of four.
    uint32_t count[kTaskCap];  // degree (| kSelfBit | kLeafAuthBit)
    uint32_t meta[kTaskCap];   // child meta
    uint32_t sid[kTaskCap];
    uint32_t scan[64];         // generic expansion / flush_probes: exclusive prefix of a round's 64 degrees
    uint64_t heads[kHeadWords];  // flush_simple: bit (w & 63) of word (w >> 6) set <=> a task's children start at work item w
};

// Wave-private output cursor, all fields wave-uniform.
//   chunked (k_expand, k_rev_expand): the wave starts on its static chunk (id = wave index); when that is full it closes it
//            (count store) and takes a dynamic chunk from the level's counter;
//   LOCAL   (k_check_local): a linear wave-private region [cur, cur + cap) of the frontier buffer.
struct WaveOutCold {  // what only the chunk switch / overflow paths need: kept in LDS, not in ~10 SGPRs for the whole kernel
    uint32_t *counts, *nchunks, *overflow;
    uint32_t nwaves, max_chunks, cap;
#if ACL_PROFILE_PHASES
    uint32_t last, prof[16];
#endif
};
struct WaveOut {
    uint4 *buf;
    uint32_t cur, fill, produced;
    WaveOutCold *cold;  // LDS
    uint32_t *lfill;    // LOCAL: the block's output cursor (LDS) -- the block's waves append to one region
    uint32_t first;     // LOCAL: first request of the unit being walked (8-byte entries hold the request's index inside the unit)
    uint32_t lcap;      // LOCAL: entries of the block's region (a kernel constant: kept here, not behind an LDS read per reservation)
};
// room for `need` (<= kChunk) consecutive entries; returns the first entry index
template <bool LOCAL>
__device__ __forceinline__ uint32_t reserve(WaveOut &wo, uint32_t need, uint32_t lane) {
    if (wo.cur == kNoSpace) return kNoSpace;
    if (LOCAL) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(wo.lfill, need);
        base = uniform(base);
        if (base + need > wo.lcap) {
            if (lane == 0) *wo.cold->overflow = 1u;
            wo.cur = kNoSpace;
            return kNoSpace;
        }
        wo.produced += need;
        return base;
    }
    if (wo.fill + need > kChunk) {
        const WaveOutCold k = *wo.cold;
        if (lane == 0) k.counts[wo.cur] = wo.fill;
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(k.nchunks, 1u);
        c = uniform(c) + k.nwaves;
        if (c >= k.max_chunks) {
            if (lane == 0) *k.overflow = 1u;
            wo.cur = kNoSpace;
            return kNoSpace;
        }
        wo.cur = c;
        wo.fill = 0;
    }
    const uint32_t base = wo.cur * kChunk + wo.fill;
    wo.fill += need;
    wo.produced += need;
    return base;
}

// Gathers address their tables as `scalar base + 32-bit byte offset` (global_load ... v_off, s[base:base+1]): one address VGPR
// per load in flight instead of a 64-bit pair.  The host guarantees every snapshot array and frontier buffer stays below
// 4 GiB (plan.cpp / alloc_frontier fail loudly beyond), which is > 1 G relationships per sorted array.
// Round 6: both say GLOBAL address space out loud.  The frontier buffers reach the walk through a two-element pointer array indexed by the level's
// parity, which loses the compiler's address-space inference: every entry load and store of the single-launch walk was a FLAT instruction (195 of
// them in the wide kernel's ISA) -- a flat access counts on lgkmcnt as well as vmcnt, so every LDS wait behind one waits for global memory too.
#if defined(__HIP_DEVICE_COMPILE__)
#define ACL_GLOBAL __attribute__((address_space(1)))
#else
#define ACL_GLOBAL  // (the host pass only parses these)
#endif
template <typename T>
__device__ __forceinline__ T gld(const T *__restrict__ base, uint32_t idx) {
    return *(const ACL_GLOBAL T *)((const ACL_GLOBAL char *)base + (uint32_t)(idx * (uint32_t)sizeof(T)));
}
template <typename T>
__device__ __forceinline__ void gst(T *__restrict__ base, uint32_t idx, const T &v) {
    *(ACL_GLOBAL T *)((ACL_GLOBAL char *)base + (uint32_t)(idx * (uint32_t)sizeof(T))) = v;
}

// Frontier entries of the single-launch walk are 8 BYTES (round 4; VERDICT r3 next #1b): the subject id and the subject key are constants of
// the REQUEST, so they live once per request in the block's LDS (s_req) and an entry keeps only what is its own:
//   x = object id[0:31) | probed[31]        y = slot[0:13) | level[13:19) | request index inside the unit[19:32)
// -- half the frontier bytes written, read back and kept in the L2, and a dwordx2 instead of a dwordx4 per lane and level.  Entries are
// decoded into the 16-byte register form (id, request, meta, subject id) right behind the load and encoded at the store: nothing else in
// process_segment knows.  The level loop (k_expand: requests of a whole batch, chunks shared by all waves) and the combine instantiations
// (entries name result CELLS, not requests) keep the 16-byte form.
template <bool E8>
__device__ __forceinline__ void put_entry(const WaveOut &wo, uint32_t idx, uint32_t id, uint32_t req, uint32_t meta, uint32_t sid) {
    if (E8) gst(reinterpret_cast<uint2 *>(wo.buf), idx, make_uint2(id | (((meta >> 19) & 1u) << 31), (meta & 0x7FFFFu) | ((req - wo.first) << 19)));
    else gst(wo.buf, idx, make_uint4(id, req, meta, sid));
}
// ... and the requests' answer bytes (has[] / err[]) of such a walk live in the block's LDS too, indexed by the request's place in the unit: the
// "is this request answered yet" read per entry and every hit / depth-error store are LDS accesses, not gathers and scattered byte stores in
// global memory (one of the ~6 vector-L1 line accesses per entry; each byte store was a 32 B write-through).  A: the array (LDS when L).
template <bool L>
__device__ __forceinline__ uint32_t ans_get(const uint8_t *a, uint32_t req, uint32_t first) {
    return L ? (uint32_t)a[req - first] : (uint32_t)gld(a, req);
}
template <bool L>
__device__ __forceinline__ void ans_set(uint8_t *a, uint32_t req, uint32_t first, uint8_t v) {
    if (L) a[req - first] = v;
    else a[req] = v;
}
__device__ __forceinline__ uint4 decode_entry8(const uint2 &v, const uint2 *sreq, uint32_t first) {
    const uint32_t rl = v.y >> 19;
    const uint2 rq = sreq[rl];  // {subject id, subject key}
    return make_uint4(v.x & 0x7FFFFFFFu, first + rl, (v.y & 0x7FFFFu) | ((v.x >> 31) << 19) | (rq.y << 20), rq.x);
}

// sorted sub-row (ids ascending; bit 31 of an edge is the leaf flag, not part of the id)
__device__ __forceinline__ bool row_contains(const uint32_t *__restrict__ edges, uint32_t lo, uint32_t hi, uint32_t key) {
    uint32_t end = hi;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if ((gld(edges, mid) & kIdMask) < key) lo = mid + 1;
        else hi = mid;
    }
    return lo < end && (gld(edges, lo) & kIdMask) == key;
}

// hashed row (plan.hpp): buckets of 4 ids; a SEEDED single-choice row keeps an id in bucket hrow_bucket(id, y) or nowhere -- ONE 16 B gather per
// membership test (round 5; rounds 1-4 placed every row two-choice: two gathers per test, and the walk's time follows the number of gather
// instructions its waves issue, profiles/r04_bucket_gather_sensitivity.txt) -- and a two-choice row (kRowTwoBit: a long row no seed was found
// for) in bucket h1 or h2: two independent gathers, never a probing chain.
// is `want` one of the four ids of a bucket?  Written as xor + min3: 4 v_xor + v_min3 + v_min + v_cmp (the obvious `p.x == want || ...`
// compiles to a chain of compares and i1 packing; the deep levels of a large batch are sensitive to VALU issue, profiles/r02_pmc_issue_breakdown.txt).
__device__ __forceinline__ bool bucket_has(const uint4 &p, uint32_t want) {
    return min(min(min(p.x ^ want, p.y ^ want), p.z ^ want), p.w ^ want) == 0u;
}
__device__ __forceinline__ bool bucket_row_contains(const uint4 *__restrict__ buckets, uint32_t b0, uint32_t y, uint32_t want) {
    const uint32_t h1 = hrow_bucket(want, y);
    bool hit = bucket_has(gld(buckets, b0 + h1), want);
    if (y & kRowTwoBit) hit = hit || bucket_has(gld(buckets, b0 + hrow_bucket2(want, y, h1)), want);
    return hit;
}

// Membership of (resource id, subject sid) in a membership-only class.  The class is stored SUBJECT-indexed:
// one hashed row of resource ids per subject.  All pending sub-checks of one request share the subject, so the
// lanes of a wave (which hold a few requests' worth of neighbouring entries) keep hitting the same descriptor
// and the same few bucket lines instead of 64 different resource rows.
__device__ __forceinline__ bool subject_row_contains(const DevGraph &g, const FwdOp &op, uint32_t id, uint32_t sid) {
    if (op.flags & OP_WILD) sid = op.K;  // `T:*`: the wildcard subject's row, whoever asks (the fast paths only take plain OP_PROBE_HASH programs)
    if (sid >= op.nrows) return false;
    const uint2 md = gld(reinterpret_cast<const uint2 *>(g.meta), op.base + sid);
    return md.y != 0u && bucket_row_contains(reinterpret_cast<const uint4 *>(g.buckets), md.x, md.y, id);
}

// Row descriptor {start, end} of (object id, sorted class op.k).  Relations with two sorted classes keep both
// descriptors in one aligned 16 B record: the state's second op re-reads the same line (an L1 hit) rather than carrying
// the record in five VGPRs across the expansions -- that cache was what kept the kernel above 96 VGPRs.
__device__ __forceinline__ uint2 row_meta(const DevGraph &g, const FwdOp &op, uint32_t id) {
    if (op.K == 2) {
        const uint4 v = gld(reinterpret_cast<const uint4 *>(g.meta), (op.base >> 1) + id);
        return op.k ? make_uint2(v.z, v.w) : make_uint2(v.x, v.y);
    }
    return gld(reinterpret_cast<const uint2 *>(g.meta), op.base + id * op.K + op.k);
}

// ---- combine programs (rewrites with `&` / `-`; plan.hpp SlotProg::combine).  A visited state with such a program gets `nleaves` fresh result
// cells (bytes of has[] / err[] behind the requests' own) and a CombineNode; its ops answer those cells exactly as a monotone walk answers a
// request; when the walk is over the nodes are evaluated deepest iteration first (resolve_node).  Everything here is compiled only into the
// CMB instantiations: the monotone kernels carry none of it.
struct CombineOut {  // (wave-uniform)
    uint4 *nodes = nullptr;                        // the region this walk appends to
    uint32_t *nnode = nullptr, *ncell = nullptr;   // its counters: LDS (single launch) or the status block (level loop)
    uint32_t node_cap = 0, cell_cap = 0, cell0 = 0;  // cell0: index of the region's first cell in has[] / err[]
    uint32_t iter = 0;                             // frontier iteration being processed
    const uint32_t *bexpr = nullptr;
};
enum : uint32_t { V_NO = 0u, V_HAS = 1u, V_ERR = 2u };
__device__ __forceinline__ uint32_t cell_value(const uint8_t *has, const uint8_t *err, uint32_t c) {
    const uint32_t h = __hip_atomic_load(has + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t e = __hip_atomic_load(err + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return h ? V_HAS : (e ? V_ERR : V_NO);
}
// value stack: two bits per value in one 64-bit register (<= kMaxLeaves values)
__device__ __forceinline__ void resolve_node(const uint4 &nd, const SlotProg *progs, const uint32_t *__restrict__ bexpr, uint8_t *has, uint8_t *err) {
    const SlotProg p = progs[nd.z & 0xFFFFu];
    const uint32_t *be = bexpr + p.combine;
    const uint32_t ntok = be[0];
    const uint32_t *tok = be + 2 + p.nleaves;
    unsigned long long st = 0;
    for (uint32_t i = 0; i < ntok; i++) {
        const uint32_t t = tok[i], kind = t & 0xFF000000u, arg = t & 0xFFFFFFu;
        uint32_t v;
        if (kind == BX_LEAF) {
            v = cell_value(has, err, nd.y + arg - 1u);
        } else if (kind == BX_LEAF_ALL) {  // a.all(b): folded from the children's cells by the member nodes (resolve_member)
            const uint32_t c = nd.y + arg - 1u;
            const uint32_t h = __hip_atomic_load(has + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), e = __hip_atomic_load(err + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            v = (e & kAllNoBit) ? V_NO : ((e & 0x7Fu) ? V_ERR : (h ? V_HAS : V_NO));
        } else if (kind == BX_EXCL) {
            const uint32_t sub = (uint32_t)st & 3u, base = (uint32_t)(st >> 2) & 3u;
            st >>= 4;
            v = base != V_HAS ? base : (sub == V_ERR ? V_ERR : (sub == V_HAS ? V_NO : V_HAS));
        } else {  // BX_OR: HAS > ERR > NO; BX_AND: NO > ERR > HAS
            bool any_has = false, any_err = false, any_no = false;
            for (uint32_t k = 0; k < arg; k++) {
                const uint32_t x = (uint32_t)st & 3u;
                st >>= 2;
                any_has |= x == V_HAS;
                any_err |= x == V_ERR;
                any_no |= x == V_NO;
            }
            v = kind == BX_OR ? (any_has ? V_HAS : (any_err ? V_ERR : V_NO)) : (any_no ? V_NO : (any_err ? V_ERR : V_HAS));
        }
        st = (st << 2) | v;
    }
    const uint32_t v = ntok ? ((uint32_t)st & 3u) : V_NO;
    if (v == V_HAS) has[nd.x] = 1;
    else if (v == V_ERR) err[nd.x] = ITEM_ERR_DEPTH;
}
// member node of an intersection arrow {x: the arrow's leaf cell, y: the child's own cell, z: iteration << 16, w: 1}: the child's verdict is folded
// into the leaf cell -- HAS: "some child holds it"; NO / ERR: bits of the err byte, set with a word-wide atomic OR (several members of one
// arrow resolve at once; the neighbouring bytes' bits stay what they are).  Resolved BEFORE the regular nodes of its iteration.
__device__ __forceinline__ void resolve_member(const uint4 &nd, uint8_t *has, uint8_t *err) {
    const uint32_t v = cell_value(has, err, nd.y);
    if (v == V_HAS) {
        has[nd.x] = 1;
        return;
    }
    uint8_t *p = err + nd.x;
    const uint32_t sh = 8u * (uint32_t)((uintptr_t)p & 3u);
    atomicOr(reinterpret_cast<unsigned int *>((uintptr_t)p & ~(uintptr_t)3), (uint32_t)(v == V_NO ? kAllNoBit : ITEM_ERR_DEPTH) << sh);
}

// Child mode: evaluate every probe of state (slot, id) at `level` for subject (key, sid) without creating tasks.
// Returns true when the state still has something to enumerate (=> it must be written to the frontier).
// `leaf_known`: the edge that produced this child carried an authoritative leaf flag (`leaf`), so the child's
// own rows need not be looked at -- a plain-subject child then touches only the request subject's rows.
__device__ __forceinline__ bool eval_child(const DevGraph &g, const SlotProg *progs, const FwdOp *ops, uint32_t slot, uint32_t level, uint32_t key,
                                           uint32_t id, uint32_t sid, bool leaf_known, bool leaf, bool &hit, bool &depth_err) {
    const SlotProg p = progs[slot];
    if (level + p.max_dlevel > kMaxLevels) depth_err = true;
    const bool userset_subject = key < g.nslots;
    // plain subjects with a known leaf flag only need the probe-only ops
    const uint32_t nops = userset_subject ? p.n_total : (leaf_known ? p.n_probe : p.n_main);
    bool push = leaf_known && !leaf;
    for (uint32_t j = 0; j < nops; j++) {
        const FwdOp op = ops[p.first + j];
        const uint32_t L = level + op.dlevel;
        if (L > kMaxLevels) continue;
        if (op.flags & OP_REFLEX) {
            if (key == op.key && id == sid) hit = true;
        } else if (op.flags & OP_PUSH_SAME) {
            push = true;
        } else if (op.flags & OP_PROBE_HASH) {
            if (key == op.key) hit |= subject_row_contains(g, op, id, sid);
        } else if (id < op.nrows) {
            const bool probe = (op.flags & OP_PROBE) && key == op.key;
            const bool look = (op.flags & OP_ENUM) && !leaf_known;
            if (probe || look) {
                const uint2 md = row_meta(g, op, id);
                if (md.y > md.x) {
                    if (probe) hit |= row_contains(g.edges, md.x, md.y, sid);
                    if (look) push = true;
                }
            }
        }
    }
    return push;
}

// Fast path of the expansion for the shape that dominates deep levels: every task's child state is "simple" -- one hashed
// probe of the request's subject (a plain subject) plus an authoritative leaf flag on the edge, nothing else -- and all
// tasks agree on (child slot, subject key).  Then there is no program to interpret, and each lane evaluates kSimpleWidth children
// per step with branch-free loads (dummy in-range addresses for inactive lanes), so the edge, descriptor and bucket
// gathers of 64 x kSimpleWidth children are in flight together instead of 64 at a time behind three dependent waits.
// Bit-for-bit the same decisions, the same output entries in the same order as the generic path.
#ifndef ACL_ISA_NO_SLOW
#define ACL_ISA_NO_SLOW 0  // 1: tools/isa_loops.py only -- the simple expansion WITHOUT its rare slow-row block, so that the static instruction counts are the main path's (never run)
#endif
#ifndef ACL_PUSH_RECHECK
#define ACL_PUSH_RECHECK 1  // the expansions look at the request's answer byte (LDS) once more before they write its children: a request answered
                            // by this step's hits -- or by another wave a moment ago -- needs none of them, and an entry not written is not read
                            // back and frees a lane of a segment of the next level (C4 253 -> 238 us; profiles/r04_push_recheck_ab.txt); 0 = A/B builds
#endif
#ifndef ACL_LOCAL_REQ
#define ACL_LOCAL_REQ 1  // the direct task lists hold the request's index INSIDE the unit: the answer-byte accesses and the entry's y word lose their subtractions (A/B builds: 0)
#endif
#ifndef ACL_DIRECT_TASKS
#define ACL_DIRECT_TASKS 1  // the deep levels' segments build their task list with the prefix sums taken in registers (process_segment); 0 = through flush_tasks (A/B builds)
#endif
#ifndef ACL_SIMPLE_WIDTH
#define ACL_SIMPLE_WIDTH 3  // children per lane and step.  Round 5: with ONE bucket per child three of them cost the registers two used to, and the wide walk moved to 12 waves per
                            // block = 6 per SIMD = room for 72 VGPRs without a spill (rounds 2-4: 2 at 64 VGPRs and 8 waves per SIMD); profiles/r05_ab_block_shape.txt
#endif
constexpr int kSimpleWidth = ACL_SIMPLE_WIDTH;  // children per lane whose buckets are in flight together
#ifndef ACL_EDGES_AHEAD
#define ACL_EDGES_AHEAD 3  // = the width (6 = both windows' edges in one trip: 1 us better on C4, 3.5 us worse on the 100 M-relationship replica)
#endif
constexpr int kEdgesAhead = ACL_EDGES_AHEAD;    // children per lane whose edges are fetched in one trip
// Returns a bit mask of the 64-task rounds it did NOT handle (the caller expands those the generic way): a round whose rows
// together exceed the head-bit window (a row of thousands of ids) or with a task at the dispatch-depth limit.
//
// Instruction count is what bounds this loop (the deep levels of a large batch keep the vector ALU busy 60 % of the launch,
// profiles/r02_pmc_issue_breakdown.txt), so everything that is a property of the TASK is done once per task, not per child:
//   - the subject's row {first bucket, count} sits in the task (DESC: fetched by the parent lane together with its has[] byte
//     and row descriptor; otherwise fetched here, one gather per 64 tasks) -- no per-child descriptor gather;
//   - depth limits are checked per task (above);
//   - work item -> task without a search: every task owns >= 1 consecutive work items, so task lanes set ONE head bit each
//     (LDS atomic or) at their first work item; a lane's task is then `tasks before this 64-item window` + the head bits at or
//     below its lane (two v_mbcnt) -- the 6-step binary search over the LDS prefix array cost ~30 VALU + 6 LDS reads per child;
//   - one output reservation per step for the W x 64 children, not one per 64.
// The steps of the simple expansion: the wave's task list holds T tasks whose children are `total` work items, t.a[j].x = first edge minus first work
// item, t.heads = one bit per task at its first work item (set by the caller's prologue).  UM: every child carries the same meta `umeta` and the
// entries are 8-byte ones (no per-task meta / subject id reads at the push).
template <bool LOCAL, bool E8, bool UM>
__device__ __forceinline__ void simple_steps(TaskLds &t, uint32_t total, WaveOut &wo, uint32_t lane, const uint32_t *__restrict__ edges, const uint4 *__restrict__ buckets,
                                             uint8_t *has, uint8_t *err, uint32_t umeta) {
    constexpr int W = kSimpleWidth;
    uint32_t before = 0;  // tasks that start before the current 64-item window
    ACL_MARK(wo, PH_PROLOGUE);
    // A step = kEdgesAhead windows of 64 children: their edges go out in one trip (one VGPR each), then the buckets follow W windows
    // at a time (8 VGPRs per child).
    constexpr int E = kEdgesAhead;
    static_assert(E % W == 0, "the bucket stage walks the fetched windows W at a time");
    for (uint32_t w0 = 0; w0 < total; w0 += 64 * E) {
        bool valid[E];
        uint32_t tj[E], edge[E];
#pragma unroll
        for (int k = 0; k < E; k++) {
            const uint32_t win = (w0 >> 6) + (uint32_t)k;
            const uint64_t hw = win < kHeadWords ? t.heads[win] : 0ull;
            const uint32_t hlo = uniform((uint32_t)hw), hhi = uniform((uint32_t)(hw >> 32));
            const uint32_t w = w0 + 64u * k + lane;
            valid[k] = w < total;
            const uint32_t wv = valid[k] ? w : total - 1;  // inactive lanes shadow the last child: every load stays in range
            // tasks starting at or before this lane's work item, minus one (beyond `total` there are no head bits: the last task)
            // (the lane's own head bit straight from the uniform mask -- inverse ballot: the mask IS the condition register; `(hw >> lane) & 1` is a 64-bit
            //  vector shift, quarter rate, plus an AND per child window)
            const uint32_t own = __builtin_amdgcn_inverse_ballot_w64(((uint64_t)hhi << 32) | hlo) ? 1u : 0u;
            const uint32_t j = before + __builtin_amdgcn_mbcnt_hi(hhi, __builtin_amdgcn_mbcnt_lo(hlo, 0u)) + own - 1u;
            before += (uint32_t)__popc(hlo) + (uint32_t)__popc(hhi);
            tj[k] = j;
            edge[k] = gld(edges, t.a[j].x + wv);
        }
        issue_fence();  // trip 1: every edge of the step
        ACL_MARK(wo, PH_EDGES);
#pragma unroll
        for (int k0 = 0; k0 < E; k0 += W) {
            if (w0 + 64u * k0 >= total) break;  // (uniform)
            uint4 p[W];
            uint32_t rq[W];
            uint32_t slow = 0u;
#pragma unroll
            for (int k = 0; k < W; k++) {
                const uint4 ta = t.a[tj[k0 + k]];  // the child's task: {edge base, first bucket, the row's y (buckets | two-choice | seed), request} in one 16-byte read
                rq[k] = ta.w;
                // (hrow_fast: full-rate arithmetic, in range for EVERY row and right for all but the slow ones -- those are redone below)
                p[k] = gld(buckets, ta.y + hrow_fast(edge[k0 + k] & kIdMask, ta.z));
                slow |= ta.z;
            }
            issue_fence();  // trip 2: the W buckets
            ACL_MARK(wo, PH_BUCKETS);
            bool hit[W], push[W];
            uint64_t pb[W];
            uint32_t pre[W], np = 0;
#pragma unroll
            for (int k = 0; k < W; k++) hit[k] = valid[k0 + k] & bucket_has(p[k], edge[k0 + k] & kIdMask);
            if (!ACL_ISA_NO_SLOW && __ballot(hrow_is_slow(slow))) {  // (rare: a child whose request's subject has a SLOW row -- two-choice, or 2^16 buckets and more: its bucket(s) by the
                                                 //  32-bit hash, one child at a time; a hit in the fast hash's bucket of such a row is still a hit: the id IS in the row)
#pragma unroll
                for (int k = 0; k < W; k++) {
                    const uint4 ta = t.a[tj[k0 + k]];
                    if (hrow_is_slow(ta.z)) {
                        const uint32_t cid = edge[k0 + k] & kIdMask, h1 = hrow_slow(cid, ta.z);
                        uint4 q = gld(buckets, ta.y + h1);
                        bool h = bucket_has(q, cid);
                        if (ta.z & kRowTwoBit) {
                            q = gld(buckets, ta.y + hrow_bucket2(cid, ta.z, h1));
                            h = h | bucket_has(q, cid);
                        }
                        hit[k] = hit[k] | (valid[k0 + k] & h);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < W; k++) {
                // (bitwise, not &&: a short-circuit puts the compare under a branch, and the compiler then waits for "maybe still pending"
                //  loads at the end of every step; depth limits were checked per task above)
                push[k] = valid[k0 + k] & !hit[k] & ((edge[k0 + k] & kLeafBit) == 0u);
            }
            // stores only after the last compare: a conditional store between two compares makes the second one's wait cover it
            // (vmcnt counts stores too, and the compiler must assume the store was not issued)
            {
                // (one branch for the step's hits, not an exec-mask round trip per child: most steps of the deep levels answer nobody)
                bool anyhit = hit[0];
#pragma unroll
                for (int k = 1; k < W; k++) anyhit = anyhit | hit[k];
                if (__ballot(anyhit)) {
#pragma unroll
                    for (int k = 0; k < W; k++)
                        if (hit[k]) ans_set<E8 && ACL_ANS_LDS>(has, rq[k], (UM && ACL_LOCAL_REQ) ? 0u : wo.first, 1);
                }
            }
#if ACL_PUSH_RECHECK
            if (E8 && ACL_ANS_LDS) {
                // a request answered by THIS step's hits (or by another wave a moment ago) needs none of its other children any more: looked at
                // once more before they are written -- an entry not written is not read back, and frees a lane of a segment of the next level
                wave_lds_fence();
                uint32_t hv[W];  // (the W answer bytes travel together: one LDS round trip, not W dependent ones)
#pragma unroll
                for (int k = 0; k < W; k++) hv[k] = ans_get<true>(has, rq[k], (UM && ACL_LOCAL_REQ) ? 0u : wo.first);
                issue_fence();
#pragma unroll
                for (int k = 0; k < W; k++) push[k] = push[k] & (hv[k] == 0u);
            }
#endif
#pragma unroll
            for (int k = 0; k < W; k++) {
                pb[k] = __ballot(push[k]);
                pre[k] = np;
                np += (uint32_t)__popcll(pb[k]);
            }
            if (np) {
                const uint32_t base = reserve<LOCAL>(wo, np, lane);
                if (base != kNoSpace) {
#pragma unroll
                    for (int k = 0; k < W; k++)
                        if (push[k]) {
                            if (UM && ACL_LOCAL_REQ)  // (the task holds the request's index inside the unit: the entry's y word without the subtraction)
                                gst(reinterpret_cast<uint2 *>(wo.buf), base + pre[k] + lanes_below(pb[k]), make_uint2((edge[k0 + k] & kIdMask) | 0x80000000u, (umeta & 0x7FFFFu) | (rq[k] << 19)));
                            else
                                put_entry<E8>(wo, base + pre[k] + lanes_below(pb[k]), edge[k0 + k] & kIdMask, rq[k], (UM ? umeta : t.meta[tj[k0 + k]]) | kProbedBit, UM ? 0u : t.sid[tj[k0 + k]]);
                        }
                }
            }
            ACL_MARK(wo, PH_PUSH);
        }
    }
}

template <bool SHARDED, bool LOCAL, bool DESC, bool E8>
__device__ __forceinline__ uint32_t flush_simple(TaskLds &t, uint32_t T, WaveOut &wo, uint32_t lane, const DevGraph &g, const SlotProg &cp, const FwdOp &pop,
                                                  uint8_t *has, uint8_t *err) {
    const uint32_t *__restrict__ edges = g.edges;
    const uint4 *__restrict__ buckets = reinterpret_cast<const uint4 *>(g.buckets);
    uint32_t skipped = 0;
    // ONE round for the whole list (<= kTaskCap = 128 tasks): every lane owns tasks `lane` and `64 + lane`.  Two rounds of 64 paid this
    // prologue -- a chain of dependent LDS round trips: counts, scan, head bits, fences -- twice per pair of segments
    // (17 % of the walk's wave-time, profiles/r02_walk_phase_breakdown.txt).
    {
        const bool mine0 = lane < T, mine1 = 64u + lane < T;
        const uint32_t cnt0 = mine0 ? (t.count[lane] & kCountMask) : 0u, cnt1 = mine1 ? (t.count[64u + lane] & kCountMask) : 0u;
        const uint32_t incl0 = wave_incl_scan(cnt0, lane);
        const uint32_t tot0 = wave_last(incl0);
        const uint32_t incl1 = wave_incl_scan(cnt1, lane) + tot0;
        const uint32_t total = wave_last(incl1);
        const uint32_t lvl0 = mine0 ? meta_level(t.meta[lane]) : 0u, lvl1 = mine1 ? meta_level(t.meta[64u + lane]) : 0u;
        const uint32_t lvl = max(lvl0, lvl1);
        if (total > 64u * kHeadWords || __ballot(lvl + pop.dlevel > kMaxLevels || lvl + cp.max_dlevel > kMaxLevels)) return ~0u;
        const uint32_t excl0 = incl0 - cnt0, excl1 = incl1 - cnt1;
        if (!DESC) {
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                const uint32_t i = h * 64u + lane;
                if (i < T) {
                    const uint32_t sidt = t.sid[i];
                    const uint2 d = gld(reinterpret_cast<const uint2 *>(g.meta), pop.base + (sidt < pop.nrows ? sidt : 0u));
                    const bool row = sidt < pop.nrows && d.y != 0u;
                    t.a[i].y = row ? d.x : 0u;
                    t.a[i].z = row ? d.y : 1u;  // (the row's y: buckets | two-choice | seed; {0, 1} = the reserved empty bucket)
                }
            }
        }
        // (t.heads is all zero here: zeroed at kernel start and behind every expansion)
        // first edge of the task minus its first work item: edge index = this + work item
        if (mine0) t.a[lane].x -= excl0;
        if (mine1) t.a[64u + lane].x -= excl1;
        if (mine0) atomicOr(reinterpret_cast<unsigned long long *>(&t.heads[excl0 >> 6]), 1ull << (excl0 & 63u));
        if (mine1) atomicOr(reinterpret_cast<unsigned long long *>(&t.heads[excl1 >> 6]), 1ull << (excl1 & 63u));
        wave_lds_fence();
        simple_steps<LOCAL, E8, false>(t, total, wo, lane, edges, buckets, has, err, 0u);
        if (lane < kHeadWords) t.heads[lane] = 0ull;  // (left zero for the next expansion: one LDS round trip less in its prologue)
        wave_lds_fence();
    }
    return skipped;
}

// Second specialised expansion: all tasks lead to one child slot whose program is at most two hashed probes followed by at
// most two enumerate ops that a child is only LOOKED at for ("does it have anything to enumerate?") -- arrow targets such as
// `namespace#view = viewer + creator + ...`.  One child per lane; the subject's row descriptors (they depend on the request,
// not on the child) are fetched together with the edge, then every bucket and every row descriptor of the child together:
// two dependent trips instead of up to six.  Same decisions and output order as the generic path.
template <bool SHARDED, bool LOCAL, bool E8>
__device__ __forceinline__ void flush_probes(TaskLds &t, uint32_t T, WaveOut &wo, uint32_t lane, const DevGraph &g, const SlotProg &cp, const FwdOp *cops,
                                             uint32_t k0, bool leafauth, uint8_t *has, uint8_t *err) {
    const uint32_t *__restrict__ edges = g.edges;
    const uint2 *__restrict__ meta2 = reinterpret_cast<const uint2 *>(g.meta);
    const uint4 *__restrict__ buckets = reinterpret_cast<const uint4 *>(g.buckets);
    const uint32_t nh = uniform(cp.n_probe), nl = leafauth ? 0u : uniform(cp.n_main - cp.n_probe);
    const FwdOp *lops = cops + cp.n_probe;
    for (uint32_t gq = 0; gq < T; gq += 64) {
        const uint32_t cnt = (gq + lane < T) ? (t.count[gq + lane] & kCountMask) : 0u;
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const uint32_t total = wave_last(incl);
        t.scan[lane] = incl - cnt;
        wave_lds_fence();
        for (uint32_t w0 = 0; w0 < total; w0 += 64) {
            const uint32_t w = w0 + lane;
            const bool valid = w < total;
            const uint32_t wv = valid ? w : total - 1;  // inactive lanes shadow the last child: every load stays in range
            uint32_t j = 0;
#pragma unroll
            for (uint32_t step = 32; step >= 1; step >>= 1)
                if (t.scan[j + step] <= wv) j += step;
            const uint32_t tj = gq + j;
            const uint32_t sid = t.sid[tj];
            const uint32_t level = meta_level(t.meta[tj]);
            // trip 1: the edge and the subject's row descriptor of every probe
            const uint32_t edge = gld(edges, t.a[tj].x + (wv - t.scan[j]));
            uint2 hd[2];
            bool hrow[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const bool hk = (uint32_t)k < nh;
                hrow[k] = hk && cops[hk ? k : 0].key == k0 && sid < cops[hk ? k : 0].nrows;
                hd[k] = gld(meta2, (hk ? cops[k].base : 0u) + (hrow[k] ? sid : 0u));
            }
            const uint32_t child = edge & kIdMask;
            // trips 2 and 3: the bucket of the first probe, then of the second (two-probe programs only), one after the other (register budget)
            bool hit = false, push = leafauth && !(edge & kLeafBit);
            auto probe = [&](const uint2 &hdk, bool hrk, uint32_t dl) {
                const bool hr = hrk && hdk.y != 0u;
                const uint32_t b0 = hr ? hdk.x : 0u, y = hr ? hdk.y : 1u;  // (no row: the reserved empty bucket)
                const uint4 bp = gld(buckets, b0 + hrow_fast(child, y));  // (in range for every row, right for the fast ones)
                bool h = bucket_has(bp, child);
                if (__ballot(hrow_is_slow(y))) {  // (rare: two-choice rows and rows of 2^16 buckets and more)
                    if (hrow_is_slow(y)) {
                        const uint32_t h1 = hrow_slow(child, y);
                        h = h || bucket_has(gld(buckets, b0 + h1), child);
                        if (y & kRowTwoBit) h = h || bucket_has(gld(buckets, b0 + hrow_bucket2(child, y, h1)), child);
                    }
                }
                if (hr && level + dl <= kMaxLevels) hit = hit || h;
            };
            if (nh > 0) probe(hd[0], hrow[0], cops[0].dlevel);
            asm volatile("" ::: "memory");  // keep the second probe's gathers behind the first's (register budget, see above)
            if (nh > 1) probe(hd[1], hrow[1], cops[1].dlevel);
            asm volatile("" ::: "memory");
            // the child's own rows are looked at only where the edge carries no authoritative leaf flag (after a patch
            // made an op distrust them): one more trip, rare
            if (nl) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const bool lk = (uint32_t)k < nl;
                    const FwdOp &lo = lops[lk ? k : 0];
                    const bool inrow = lk && child < lo.nrows;
                    const uint2 md = gld(meta2, lk ? lo.base + (inrow ? child : 0u) * lo.K + lo.k : 0u);
                    if (inrow && md.y > md.x && level + lo.dlevel <= kMaxLevels) push = true;
                }
            }
            const uint32_t req = t.a[tj].w, meta = t.meta[tj];
            hit = hit && valid;
            push = push && valid;
            if (hit) {
                ans_set<E8 && ACL_ANS_LDS>(has, req, wo.first, 1);
                push = false;
            } else if (valid && level + cp.max_dlevel > kMaxLevels) {
                ans_set<E8 && ACL_ANS_LDS>(err, req, wo.first, ITEM_ERR_DEPTH);
            }
#if ACL_PUSH_RECHECK
            if (E8 && ACL_ANS_LDS) {  // (as in flush_simple: a request answered meanwhile needs none of its other children)
                wave_lds_fence();
                push = push && ans_get<true>(has, req, wo.first) == 0u;
            }
#endif
            const uint64_t b = __ballot(push);
            if (b) {
                const uint32_t base = reserve<LOCAL>(wo, (uint32_t)__popcll(b), lane);
                if (push && base != kNoSpace) put_entry<E8>(wo, base + lanes_below(b), child, req, meta | kProbedBit, sid);
            }
        }
        wave_lds_fence();
    }
}

// Expand the first T tasks of the wave's LDS list.  INLINE: children are probed here and only the ones with
// remaining enumeration work are written (forward Check); otherwise every child is written (reverse walk).
// wave-cooperative append of the flagged lanes' entries to the shard's export buffer (one atomic per call and destination).
// sh.by_dest: one buffer per destination shard (`dest` = owner of the entry's slot) for an all-to-all exchange; otherwise
// one buffer that every shard receives (all-gather; also the reverse walk's broadcast of visited states).
__device__ __forceinline__ void export_entries(bool xport, const uint4 &e, uint32_t dest, uint32_t lane, const DevShard &sh) {
    uint64_t todo = __ballot(xport);
    if (!todo) return;
    if (!sh.by_dest) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(sh.exp_count, (uint32_t)__popcll(todo));
        const uint32_t at = uniform(base) + lanes_below(todo);
        if (xport && at < sh.cap) sh.exp[at] = e;
        return;
    }
    while (todo) {
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)dest, (int)(__ffsll((unsigned long long)todo) - 1));
        const bool mine = xport && dest == d;
        const uint64_t m = __ballot(mine);
        todo &= ~m;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(sh.exp_count + 1 + d, (uint32_t)__popcll(m));
        const uint32_t at = uniform(base) + lanes_below(m);
        if (mine && at < sh.cap) sh.exp[(size_t)d * (sh.stride ? sh.stride : sh.cap) + at] = e;
    }
}

template <bool INLINE, bool SHARDED, bool LOCAL, bool DESC = false, bool CMB = false>  // DESC: the tasks carry the subject's hashed row (TaskLds b0 / nb); CMB: child slots may hold combine programs
__device__ __forceinline__ void flush_tasks(TaskLds &t, uint32_t T, WaveOut &wo, uint32_t lane, const DevGraph &g, const SlotProg *progs,
                                            const FwdOp *ops, const uint32_t *__restrict__ edges, uint8_t *has, uint8_t *err, const DevShard &sh,
                                            bool same = false /* the caller made every task from ONE op: child slot, key and flags agree */,
                                            const CombineOut &co = CombineOut() /* CMB: where the children of an intersection arrow get their cells and member nodes */) {
    constexpr bool E8 = ACL_ENTRY8 && INLINE && LOCAL && !CMB;  // (8-byte entries: the single-launch walk's monotone instantiations)
    uint32_t only = ~0u;  // rounds of 64 tasks left for the generic loop
    wave_lds_fence();
    if (INLINE) {  // all tasks lead to the same "simple" child state?  (one hashed probe + authoritative leaf flags, plain subject)
        const uint32_t m0 = t.meta[0];
        const uint32_t cs = uniform(meta_slot(m0)), k0 = uniform(meta_key(m0));
        const SlotProg cp = progs[cs];
        bool ok = cp.n_probe == 1 && k0 >= g.nslots && (!SHARDED || cp.owner == sh.rank);
        FwdOp pop{};
        if (ok) {
            pop = ops[cp.first];
            ok = pop.flags == OP_PROBE_HASH && pop.key == k0;
        }
        bool agree = true;
        if (same) {
            agree = (t.count[0] & (kLeafAuthBit | kSelfBit)) == kLeafAuthBit;
        } else {
            for (uint32_t i = lane; i < T; i += 64) {
                const uint32_t mi = t.meta[i], ci = t.count[i];
                agree = agree && meta_slot(mi) == cs && meta_key(mi) == k0 && (ci & kLeafAuthBit) && !(ci & (kSelfBit | kAllBit));
            }
        }
        if (ok && !__ballot(!agree)) {
            only = flush_simple<SHARDED, LOCAL, DESC, E8>(t, T, wo, lane, g, cp, pop, has, err);
            if (!only) return;
        } else
        // second shape: <= 2 hashed probes + <= 2 enumerate ops that are only looked at; uniform slot, key and leaf authority
        if (k0 >= g.nslots && (!SHARDED || cp.owner == sh.rank) && cp.n_probe <= 2 && cp.n_main - cp.n_probe <= 2 && !(CMB && cp.combine)) {
            const FwdOp *cops = ops + cp.first;
            bool shape = true;
            for (uint32_t q = 0; q < cp.n_main; q++) {
                const uint32_t fl = cops[q].flags;
                shape = shape && (q < cp.n_probe ? fl == OP_PROBE_HASH : ((fl & OP_ENUM) && !(fl & (OP_PUSH_SAME | OP_REFLEX | OP_PROBE_HASH)) && cops[q].K != 2));
            }
            const bool la0 = (t.count[0] & kLeafAuthBit) != 0;
            bool agree2 = true;
            for (uint32_t i = lane; i < T; i += 64) {
                const uint32_t mi = t.meta[i], ci = t.count[i];
                agree2 = agree2 && meta_slot(mi) == cs && meta_key(mi) == k0 && ((ci & kLeafAuthBit) != 0) == la0 && !(ci & (kSelfBit | kAllBit));
            }
            if (shape && !__ballot(!agree2)) {
                flush_probes<SHARDED, LOCAL, E8>(t, T, wo, lane, g, cp, cops, k0, la0, has, err);
                return;
            }
        }
    }
    for (uint32_t gq = 0; gq < T; gq += 64) {
        if (!(only & (1u << (gq >> 6)))) continue;
        const uint32_t cnt = (gq + lane < T) ? (t.count[gq + lane] & kCountMask) : 0u;
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const uint32_t total = wave_last(incl);
        t.scan[lane] = incl - cnt;
        wave_lds_fence();
        for (uint32_t w0 = 0; w0 < total; w0 += 64) {
            const uint32_t w = w0 + lane;
            bool push = false, xport = false, live = false, isall = false;
            uint4 e = make_uint4(0, 0, 0, 0);
            uint32_t c = 0, edge = 0, child = 0, s = 0, j = 0;
            if (w < total) {
                // largest j with scan[j] <= w
#pragma unroll
                for (uint32_t step = 32; step >= 1; step >>= 1)
                    if (t.scan[j + step] <= w) j += step;
                const uint32_t tj = gq + j;
                c = t.count[tj];
                s = t.a[tj].x;
                edge = (c & kSelfBit) ? s : gld(edges, s + (w - t.scan[j]));
                child = INLINE ? (edge & kIdMask) : edge;
                e = make_uint4(child, t.a[tj].w, t.meta[tj], t.sid[tj]);
                push = live = true;
                isall = CMB && (c & kAllBit);
            }
            if (CMB) {
                // children of an intersection arrow a.all(b): a result cell of its own for every child, and a member node that folds the child's
                // verdict into the arrow's leaf cell (resolve_member) -- one reservation per wave and window
                const uint64_t ab = __ballot(isall);
                if (ab) {
                    const uint32_t k = (uint32_t)__popcll(ab);
                    uint32_t c0 = 0, n0 = 0;
                    if (lane == 0) {
                        c0 = atomicAdd(co.ncell, k);
                        n0 = atomicAdd(co.nnode, k);
                    }
                    c0 = uniform(c0);
                    n0 = uniform(n0);
                    if (c0 + k > co.cell_cap || n0 + k > co.node_cap) {  // out of cells / nodes: as where a combine state is visited (process_segment)
                        if (LOCAL) {
                            if (lane == 0) *wo.cold->overflow = 1u;
                            wo.cur = kNoSpace;
                        } else if (lane == 0) {
                            *wo.cold->overflow = SHARDED ? kOverflowPools : 3u;
                        }
                        if (isall) push = live = false;
                    } else if (isall) {
                        const uint32_t r = lanes_below(ab), cell = co.cell0 + c0 + r;
                        has[cell] = 0;
                        err[cell] = ITEM_ERR_NONE;
                        co.nodes[n0 + r] = make_uint4(e.y, cell, co.iter << 16, 1u);
                        e.y = cell;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the zeroes are acknowledged before a probe of the child stores a hit)
                }
            }
            if (live) {
                if (INLINE && SHARDED && progs[meta_slot(e.z)].owner != sh.rank) {
                    // the child's rows live on another shard: it leaves unprobed and is evaluated by its owner
                    push = false;
                    xport = true;
                } else if (INLINE && CMB && progs[meta_slot(e.z)].combine) {
                    // a combine child is visited as a state of its own, unprobed: its probes answer leaf cells that only exist once it is
                } else if (INLINE) {
                    bool hit = false, derr = false;
                    push = eval_child(g, progs, ops, meta_slot(e.z), meta_level(e.z), meta_key(e.z), child, e.w, (c & kLeafAuthBit) != 0,
                                      (edge & kLeafBit) != 0, hit, derr);
                    if (hit) {
                        ans_set<E8 && ACL_ANS_LDS>(has, e.y, wo.first, 1);
                        push = false;
                    } else if (derr) {
                        ans_set<E8 && ACL_ANS_LDS>(err, e.y, wo.first, ITEM_ERR_DEPTH);
                    }
                    e.z |= kProbedBit;
                    if (ACL_PERTURB == 3) {
                        const FwdOp pop = ops[progs[meta_slot(e.z)].first];
                        if (pop.flags & OP_PROBE_HASH) { const bool x = subject_row_contains(g, pop, child ^ 1u, e.w); ACL_KEEP((uint32_t)x); }
                    }
                    if (ACL_PERTURB == 5 && !(c & kSelfBit)) { const uint32_t x = edges[s + ((w - t.scan[j]) ^ 1u)]; ACL_KEEP(x); }
                    if (ACL_PERTURB == 6) {
                        uint32_t x = child;
#pragma unroll
                        for (int q = 0; q < 20; q++) x = x * 0x9E3779B1u + (x >> 7);
                        ACL_KEEP(x);
                    }
                    if (ACL_PERTURB == 7) { const uint4 x = reinterpret_cast<const uint4 *>(g.buckets)[(child * 0x9E3779B1u) >> 12]; ACL_KEEP(x.x); }
                    if (ACL_PERTURB == 8) {
                        uint32_t x = 0;
#pragma unroll
                        for (int q = 0; q < 6; q++) x += t.meta[(gq + j + q * 7 + x) & (kTaskCap - 1)];
                        ACL_KEEP(x);
                    }
                }
            }
#if ACL_PUSH_RECHECK
            if (E8 && ACL_ANS_LDS && INLINE) {  // (as in flush_simple: a request answered meanwhile needs none of its other children)
                wave_lds_fence();
                push = push && ans_get<true>(has, e.y, wo.first) == 0u;
            }
#endif
            const uint64_t b = __ballot(push);
            if (b) {
                const uint32_t base = reserve<LOCAL>(wo, (uint32_t)__popcll(b), lane);
                if (push && base != kNoSpace) put_entry<E8>(wo, base + lanes_below(b), e.x, e.y, e.z, e.w);
            }
            if (INLINE && SHARDED) export_entries(xport, e, xport ? progs[meta_slot(e.z)].owner : 0u, lane, sh);
        }
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------ seed
// items: acl_item_t (16 B): x = rtype | perm << 16, y = resource id, z = stype | srel << 16, w = subject id
__global__ __launch_bounds__(256) void k_seed(DevGraph g, DevFrontier f, const uint4 *__restrict__ items, uint32_t n, uint8_t *has, uint8_t *err,
                                              DevShard sh) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t need = (n + kChunk - 1) / kChunk;       // chunks holding seeds: ids [0, need)
    const uint32_t readable = max(need, f.nwaves);          // the reader scans every static chunk
    // status block = nchunks[kLevelSlots] | any[kLevelSlots] | overflow | export count, contiguous from f.nchunks:
    // reset here (no separate memset launch); slot 0 describes the seeds
    if (i < kStatusWords) f.nchunks[i] = i == 0 ? (need > f.nwaves ? need - f.nwaves : 0u) : (i == kLevelSlots ? 1u : 0u);
    if (i < readable) f.counts[0][i] = i < need ? min(kChunk, n - i * kChunk) : 0u;
    if (i >= n) return;
    uint4 it = items[i];
    uint32_t rtype = it.x & 0xFFFFu, perm = it.x >> 16, stype = it.z & 0xFFFFu, srel = it.z >> 16;
    bool ok = rtype < g.ntypes && stype < g.ntypes && perm < g.type_nmembers[rtype < g.ntypes ? rtype : 0] &&
              (srel == 0xFFFFu || srel < g.type_nmembers[stype < g.ntypes ? stype : 0]);
    has[i] = 0;
    err[i] = ok ? ITEM_ERR_NONE : ITEM_ERR_INVALID;
    uint32_t meta = kDeadMeta;
    if (ok) {
        uint32_t slot = g.type_slot_base[rtype] + perm;
        uint32_t key = srel == 0xFFFFu ? g.nslots + stype : g.type_slot_base[stype] + srel;
        meta = make_meta(slot, 1u, key);
        if (sh.world > 1 && g.progs[slot].owner != sh.rank) meta = kDeadMeta;  // seeded by the shard that owns the resource type
    }
    f.buf[0][i] = make_uint4(it.y, i, meta, it.w);
}

// reverse-walk seeds: lookup request i starts from subject sids[i]; entry meta = subject key (dist 0 => program rseeds[key]).
// Also resets the status block (slot 0 describes the seeds), like k_seed.
__global__ __launch_bounds__(256) void k_rev_seed(DevFrontier f, const uint32_t *__restrict__ sids, uint32_t n, uint32_t key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t need = (n + kChunk - 1) / kChunk;
    const uint32_t readable = max(need, f.nwaves);
    if (i < kStatusWords) f.nchunks[i] = i == 0 ? (need > f.nwaves ? need - f.nwaves : 0u) : (i == kLevelSlots ? (n ? 1u : 0u) : 0u);
    if (i < readable) f.counts[0][i] = i < need ? min(kChunk, n - i * kChunk) : 0u;
    if (i < n) f.buf[0][i] = make_uint4(sids[i], i, key, 0);
}

// ---------------------------------------------------------------- expand
// One 64-entry segment of the frontier: every lane holds one pending sub-check.  Shared by the level-synchronous kernel
// (k_expand: segments come from the chunked global frontier) and the single-launch kernel (k_check_local: segments
// come from the wave's private region).  `next` lets the simple-parent fast path pull the following segment in early:
//   bool peek(uint4 &e, bool &valid)  loads the next segment's entries if there is one (not consumed yet)
//   void take()                       consumes it
template <bool SHARDED, bool LOCAL, bool CMB, typename Next, bool SEEDS = false /* the entries are a batch's seeds: never probed, so never "simple parents" -- the fast path is not compiled in */>
__device__ __forceinline__ void process_segment(const uint4 &e, bool valid, Next &next, TaskLds &t, WaveOut &wo, uint32_t lane, const DevGraph &g,
                                                const SlotProg *progs, const FwdOp *ops, uint8_t *has, uint8_t *err, const DevShard &sh,
                                                const CombineOut &co = CombineOut()) {
    constexpr bool E8 = ACL_ENTRY8 && LOCAL && !CMB;  // 8-byte entries + answers in LDS (put_entry, ans_get)
    const uint32_t id = e.x, req = e.y, meta = e.z;
    // ---- fast path: every entry of the segment is a "simple parent" -- probes already done by its own parent (kProbedBit),
    // plain subject, and a program whose only remaining op enumerates one sorted row.  No interpreter: the has[] read, the
    // row-descriptor gather and the subject's row of the child's probe are issued together (branch-free), for this segment (A)
    // and the wave's next one (B) at once.  The slots may differ from lane to lane (level 2 of a pod check holds `namespace#view`
    // and `group#member` states side by side): each lane reads ITS program from the LDS copy; a segment of one slot (the deep
    // levels) reads it once, through the scalar path.
    if (!__ballot(valid)) return;
    if (!SEEDS) {
        const uint64_t vb = __ballot(valid);
        // B is loaded right behind A: their entries arrive in one trip
        bool validB = false;
        uint4 eB = make_uint4(0, 0, kDeadMeta, 0);
        const bool haveB = next.peek(eB, validB);
        issue_fence();
        ACL_MARK(wo, PH_ENTRIES);
        struct LaneOp {  // the lane's one remaining op + what its tasks need to know about the child state
            uint32_t flags, dlevel, base, nrows, Kk, key, maxd;  // Kk = K | k << 16
            uint32_t cbase, cnrows, ckey;                        // child's hashed probe (cnrows == 0: the child is not "one hashed probe")
            uint32_t cdl, cmaxd;                                 // ... its dispatch-depth offset and the child program's deepest inlined state
        };
        auto lane_op = [&](uint32_t m, LaneOp &L) -> bool {  // false: not a simple parent
            if (m == kDeadMeta || !(m & kProbedBit) || meta_key(m) < g.nslots) return false;
            const SlotProg sp = progs[meta_slot(m)];
            if (sp.n_main != sp.n_probe + 1 || (CMB && sp.combine)) return false;
            const FwdOp o = ops[sp.first + sp.n_probe];
            if (!(o.flags & OP_ENUM) || (o.flags & (OP_PUSH_SAME | OP_REFLEX | OP_PROBE_HASH))) return false;
            L.flags = o.flags; L.dlevel = o.dlevel; L.base = o.base; L.nrows = o.nrows; L.Kk = o.K | (o.k << 16); L.key = o.key; L.maxd = sp.max_dlevel;
            const SlotProg cp = progs[o.key];
            L.cbase = 0; L.cnrows = 0; L.ckey = 0; L.cdl = 0; L.cmaxd = cp.max_dlevel;
            if (cp.n_probe == 1) {
                const FwdOp co = ops[cp.first];
                if (co.flags == OP_PROBE_HASH) { L.cbase = co.base; L.cnrows = co.nrows; L.ckey = co.key; L.cdl = co.dlevel; }
            }
            return true;
        };
        const uint32_t m0 = (uint32_t)__builtin_amdgcn_readlane((int)meta, (int)(__ffsll((unsigned long long)vb) - 1));
        LaneOp LA{}, LB{};
        bool simple;
        const uint64_t vbB = haveB ? __ballot(validB) : 0ull;
        const bool oneslot = !__ballot(valid && meta != m0) && !(vbB && __ballot(validB && eB.z != m0));  // (same slot, level, key: the deep levels)
        if (oneslot) {
#if ACL_DIRECT_TASKS
            // ---- the deep levels' shape, without the task list's round trips (round 5): ONE slot, level and subject key for both segments, children
            // that are "one hashed probe + authoritative leaf flags" (what flush_simple takes).  Decided and run HERE, while the program's fields are
            // still scalars.  The lane that creates a task already holds its degree, so the prefix sums run over the creating lanes' registers
            // (compaction keeps the order: the prefix over lanes IS the prefix over tasks), each lane writes its task with the edge base already
            // rebased and sets its head bit itself, and every child carries the same meta: no count / meta / subject-id stores, no read-back of the
            // counts, no read-modify-write of the edge bases -- the prologue's chain of dependent LDS trips (14 % of the walk's wave-time,
            // profiles/r05_ab_split_units.txt) shrinks to one fence.
            LaneOp LD{};  // (a copy of its own: nothing of it is live behind this block, so the general path's LA / LB are not carried -- spilled -- across it)
            if (E8 && !(g.walk_flags & kWalkNoDirect) && lane_op(m0, LD)) {
                // (the program's fields come out of the LDS copy in VGPRs, uniform or not: pinned into SGPRs here -- a dozen VGPRs held across the
                //  gathers otherwise, and the walk runs at exactly 64)
                const uint32_t lv = meta_level(m0), Lv = lv + uniform(LD.dlevel);
                const uint32_t d_cnrows = uniform(LD.cnrows), d_cbase = uniform(LD.cbase), d_base = uniform(LD.base), d_nrows = uniform(LD.nrows), d_Kk = uniform(LD.Kk), d_key = uniform(LD.key);
                if (d_cnrows != 0u && uniform(LD.ckey) == meta_key(m0) && (uniform(LD.flags) & OP_LEAFBIT) != 0u && Lv + 1u <= kMaxLevels && Lv + 1u + uniform(LD.cdl) <= kMaxLevels &&
                    Lv + 1u + uniform(LD.cmaxd) <= kMaxLevels && lv + uniform(LD.maxd) <= kMaxLevels) {
                    const bool dpair = vbB != 0;
                    if (dpair) next.take();
                    const uint2 *__restrict__ dmeta2 = reinterpret_cast<const uint2 *>(g.meta);
                    const uint32_t rK = d_Kk & 0xFFFFu, rk = d_Kk >> 16;
                    const bool vB = dpair && validB;
                    // all six gathers in flight together
                    const uint32_t hvA = ans_get<true>(has, valid ? req : wo.first, wo.first), hvB = ans_get<true>(has, vB ? eB.y : wo.first, wo.first);
                    const bool inA = valid && id < d_nrows, inB = vB && eB.x < d_nrows;
                    const uint2 mdA = gld(dmeta2, d_base + (inA ? id * rK + rk : 0u)), mdB = gld(dmeta2, d_base + (inB ? eB.x * rK + rk : 0u));
                    const bool okA = valid && e.w < d_cnrows, okB = vB && eB.w < d_cnrows;
                    uint2 sdA = gld(dmeta2, d_cbase + (okA ? e.w : 0u)), sdB = gld(dmeta2, d_cbase + (okB ? eB.w : 0u));
                    issue_fence();
                    ACL_MARK(wo, PH_GATHERS);
                    if (!(okA && sdA.y != 0u)) sdA = make_uint2(0u, 1u);  // no row: the reserved empty bucket
                    if (!(okB && sdB.y != 0u)) sdB = make_uint2(0u, 1u);
                    const uint32_t degA = (valid && !hvA && inA && mdA.y > mdA.x) ? mdA.y - mdA.x : 0u;
                    const uint32_t degB = (vB && !hvB && inB && mdB.y > mdB.x) ? mdB.y - mdB.x : 0u;
                    const uint32_t inclA = wave_incl_scan(degA, lane), totA = wave_last(inclA);
                    const uint32_t inclB = wave_incl_scan(degB, lane) + totA, total = wave_last(inclB);
                    if (total > 64u * kHeadWords || __ballot(degA > kMaxRow || degB > kMaxRow)) {
                        // more children than the head-bit window maps (an average fan-out beyond 16), or a row beyond the enumeration limit: this form has
                        // no second round (one would have to keep the pair's registers across the steps -- the walk runs at exactly 64 VGPRs).  The batch is
                        // redone on the level loop and the host switches the form off for this snapshot (kOverflowDirect; a row beyond kMaxRow fails
                        // the call as it does on every path).
                        if (lane == 0) *wo.cold->overflow = __ballot(degA > kMaxRow || degB > kMaxRow) ? 2u : kOverflowDirect;
                        wo.cur = kNoSpace;
                        return;
                    }
                    {
                        if (!total) return;
                        const uint64_t bA = __ballot(degA != 0u), bB = __ballot(degB != 0u);
                        if (degA) {
                            const uint32_t ex = inclA - degA;
                            t.a[lanes_below(bA)] = make_uint4(mdA.x - ex, sdA.x, sdA.y, ACL_LOCAL_REQ ? req - wo.first : req);
                            atomicOr(reinterpret_cast<unsigned long long *>(&t.heads[ex >> 6]), 1ull << (ex & 63u));
                        }
                        if (degB) {
                            const uint32_t ex = inclB - degB;
                            t.a[(uint32_t)__popcll(bA) + lanes_below(bB)] = make_uint4(mdB.x - ex, sdB.x, sdB.y, ACL_LOCAL_REQ ? eB.y - wo.first : eB.y);
                            atomicOr(reinterpret_cast<unsigned long long *>(&t.heads[ex >> 6]), 1ull << (ex & 63u));
                        }
                        wave_lds_fence();
                        ACL_MARK(wo, PH_TASKS);
                        simple_steps<LOCAL, true, true>(t, total, wo, lane, g.edges, reinterpret_cast<const uint4 *>(g.buckets), has, err, make_meta(d_key, Lv + 1u, meta_key(m0)));
                        if (lane < kHeadWords) t.heads[lane] = 0ull;
                        wave_lds_fence();
                        ACL_MARK(wo, PH_PUSH);
                        return;
                    }
                }
            }
#endif
            simple = lane_op(m0, LA);  // wave-uniform argument
            LB = LA;
        } else {
            simple = !__ballot(valid && !lane_op(meta, LA));
        }
        if (simple) {
            bool pairB = vbB != 0;
            if (pairB && !oneslot) pairB = !__ballot(validB && !lane_op(eB.z, LB));
            if (pairB) next.take();
            else validB = false;
            const uint2 *__restrict__ meta2 = reinterpret_cast<const uint2 *>(g.meta);
            auto subj_desc = [&](const uint4 &se, bool sv, const LaneOp &L) -> uint2 {
                const bool ok = sv && meta_key(se.z) == L.ckey && se.w < L.cnrows;
                const uint2 d = gld(meta2, L.cbase + (ok ? se.w : 0u));
                return (ok && d.y != 0u) ? d : make_uint2(0u, 1u);  // {first bucket, the row's y}; no row: the reserved empty bucket
            };
            auto row_desc = [&](uint32_t rid, bool in, const LaneOp &L) -> uint2 { return gld(meta2, L.base + (in ? rid * (L.Kk & 0xFFFFu) + (L.Kk >> 16) : 0u)); };
            // tasks of one simple segment -> LDS task slots [Tb, Tb + n); returns n
            auto seg_tasks = [&](const uint4 &se, bool sv, uint32_t hv, uint2 md, uint2 sd, bool inrow, const LaneOp &L, uint32_t Tb) -> uint32_t {
                const bool act = sv && !hv;
                const uint32_t lv = meta_level(se.z), Lv = lv + L.dlevel;
                bool derr = act && lv + L.maxd > kMaxLevels, want = false;
                if (act && Lv <= kMaxLevels && inrow && md.y > md.x) {
                    if (Lv + 1 > kMaxLevels) derr = true;
                    else if (md.y - md.x > kMaxRow) *wo.cold->overflow = 2u;
                    else want = true;
                }
                if (derr) ans_set<E8 && ACL_ANS_LDS>(err, se.y, wo.first, ITEM_ERR_DEPTH);
                const uint64_t b = __ballot(want);
                if (want) {
                    const uint32_t q = Tb + lanes_below(b);
                    t.a[q] = make_uint4(md.x, sd.x, sd.y, se.y);
                    t.count[q] = (md.y - md.x) | ((L.flags & OP_LEAFBIT) ? kLeafAuthBit : 0u);
                    t.meta[q] = make_meta(L.key, Lv + 1, meta_key(se.z));
                    t.sid[q] = se.w;
                }
                return (uint32_t)__popcll(b);
            };
            // all six gathers in flight together
            const uint32_t hvA = ans_get<E8 && ACL_ANS_LDS>(has, valid ? req : ((E8 && ACL_ANS_LDS) ? wo.first : 0u), wo.first);
            const bool inA = valid && id < LA.nrows;
            const uint2 mdA = row_desc(id, inA, LA);
            const uint2 sdA = subj_desc(e, valid, LA);
            uint32_t hvB = 0;
            uint2 mdB = make_uint2(0, 0), sdB = make_uint2(0, 1);
            const bool inB = validB && eB.x < LB.nrows;
            if (pairB) {
                hvB = ans_get<E8 && ACL_ANS_LDS>(has, validB ? eB.y : ((E8 && ACL_ANS_LDS) ? wo.first : 0u), wo.first);
                mdB = row_desc(eB.x, inB, LB);
                sdB = subj_desc(eB, validB, LB);
            }
            issue_fence();
            ACL_MARK(wo, PH_GATHERS);
#if ACL_PROFILE_PHASES  // (instrumented builds: entries the deep levels read, and how many of them belong to requests that were answered meanwhile)
            {
                const uint32_t nA = (uint32_t)__popcll(__ballot(valid)), dA = (uint32_t)__popcll(__ballot(valid && hvA != 0u));
                const uint32_t nB = pairB ? (uint32_t)__popcll(__ballot(validB)) : 0u, dB = pairB ? (uint32_t)__popcll(__ballot(validB && hvB != 0u)) : 0u;
                if (lane == 0) {
                    wo.cold->prof[12] += nA + nB;
                    wo.cold->prof[13] += dA + dB;
                    wo.cold->prof[14] += 1u + (pairB ? 1u : 0u);                                  // segments walked
                    wo.cold->prof[15] += (nA + nB - dA - dB <= 64u && pairB) ? 1u : 0u;            // pairs whose live entries would fit ONE segment
                }
            }
#endif
            uint32_t T = seg_tasks(e, valid, hvA, mdA, sdA, inA, LA, 0u);
            if (pairB) T += seg_tasks(eB, validB, hvB, mdB, sdB, inB, LB, T);
            ACL_MARK(wo, PH_TASKS);
            if (T) flush_tasks<true, SHARDED, LOCAL, true, CMB>(t, T, wo, lane, g, progs, ops, g.edges, has, err, sh, oneslot);
            ACL_MARK(wo, PH_PUSH);
            return;
        }
    }
    // ---- generic path: interpret the state's program.  Tasks are only RECORDED while the ops run (one segment of the LDS
    // list per enumerating op) and expanded afterwards, segment by segment, when the interpreter's per-lane state is dead:
    // the expansions (flush_simple / flush_probes) are the register-hungry part and no longer stack on top of it.
    // (Per-op segments keep the tasks of one flush uniform in child slot, which is what the fast paths need.)
    constexpr uint32_t kMaxSeg = 4;
    uint4 ee = e;
    uint32_t jstart = 0;
    bool hit = false;
    uint32_t cbase = 0;  // CMB: first leaf cell of this lane's combine state (allocated on the first pass over the entry)
    for (;;) {
        // (re)derive everything from the entry: nothing but `ee`, `hit` and `jstart` lives across the expansions below
        asm volatile("" : "+v"(ee.x), "+v"(ee.y), "+v"(ee.z), "+v"(ee.w));
        const uint32_t id = ee.x, req = ee.y, meta = ee.z, sid = ee.w;
        bool active = valid && meta != kDeadMeta;
        if (active && (hit || ans_get<E8 && ACL_ANS_LDS>(has, req, wo.first))) active = false;  // request already answered HAS: drop its pending work
        const uint32_t slot = meta_slot(meta), level = meta_level(meta), key = meta_key(meta);
        const bool probed = meta & kProbedBit;  // the parent already ran this state's probes
        SlotProg p{};
        if (active) p = progs[slot];
        bool cmb = false;
        if (CMB) {
            cmb = active && p.combine != 0;
            if (jstart == 0) {  // first pass over the entry: leaf cells, the node, the leaves' depth errors
                bool over = false;
                if (cmb) {
                    const uint32_t c0 = atomicAdd(co.ncell, p.nleaves), ni = atomicAdd(co.nnode, 1u);
                    over = c0 + p.nleaves > co.cell_cap || ni >= co.node_cap;
                    if (!over) {
                        cbase = co.cell0 + c0;
                        for (uint32_t k = 0; k < p.nleaves; k++) {
                            has[cbase + k] = 0;
                            err[cbase + k] = ITEM_ERR_NONE;
                        }
                        co.nodes[ni] = make_uint4(req, cbase, slot | (co.iter << 16), 0u);
                    }
                }
                if (__ballot(over)) {  // out of cells / nodes: the pass is redone elsewhere (single launch -> level loop) or fails (level loop)
                    if (LOCAL) {
                        if (lane == 0) *wo.cold->overflow = 1u;
                        wo.cur = kNoSpace;
                    } else if (lane == 0) {
                        *wo.cold->overflow = SHARDED ? kOverflowPools : 3u;
                    }
                    if (over) active = cmb = false;
                }
                // (the zeroes above are acknowledged before any hit is stored into a cell, by this lane or by the lanes that expand its tasks)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (cmb) {
                    const uint32_t *be = co.bexpr + p.combine + 1;  // deepest inlined dispatch offset per leaf (0 = the direct ops)
                    for (uint32_t k = 0; k <= p.nleaves; k++)
                        if (level + be[k] > kMaxLevels) err[k ? cbase + k - 1u : req] = ITEM_ERR_DEPTH;
                }
            }
        }
        const uint32_t j0 = probed ? p.n_probe : 0u;
        const uint32_t j1 = active ? ((probed || key >= g.nslots) ? p.n_main : p.n_total) : 0u;
        bool depth_err = active && !cmb && level + p.max_dlevel > kMaxLevels;
        uint32_t T = 0, nseg = 0, seg_end[kMaxSeg] = {0, 0, 0, 0};
            const uint32_t maxops = uniform(wave_max(j1 > j0 ? j1 - j0 : 0u));
        uint32_t jj = jstart;
        for (; jj < maxops; jj++) {
            const uint32_t j = j0 + jj;
            bool want = false;
            uint32_t tstart = 0, tcount = 0, tmeta = 0, tcell = req;
            if (j < j1) {
                const FwdOp op = ops[p.first + j];
                const uint32_t L = level + op.dlevel;
                const bool leafop = CMB && op.leaf != 0;  // the op answers one of the state's leaf cells, not the entry's own cell
                if (leafop) tcell = cbase + op.leaf - 1u;
                if (L <= kMaxLevels) {
                    bool h = false, derr = false;  // this op's probe hit / it would dispatch beyond the depth limit
                    if (op.flags & OP_REFLEX) {
                        h = key == op.key && id == sid;
                    } else if (op.flags & OP_PUSH_SAME) {
                        if (L + 1 > kMaxLevels) derr = true;
                        else {
                            want = true;
                            tstart = id;
                            tcount = 1u | kSelfBit;
                            tmeta = make_meta(op.key, L + 1, key);
                        }
                    } else if (op.flags & OP_PROBE_HASH) {
                        if (key == op.key) h = subject_row_contains(g, op, id, sid);
                    } else if (id < op.nrows) {
                        const uint2 md = row_meta(g, op, id);
                        if (md.y > md.x) {
                            if ((op.flags & OP_PROBE) && key == op.key) h = row_contains(g.edges, md.x, md.y, sid);
                            if (op.flags & OP_ENUM) {
                                if (L + 1 > kMaxLevels) derr = true;
                                else if (md.y - md.x > kMaxRow) *wo.cold->overflow = 2u;
                                else if (!(leafop ? h : (hit || h))) {
                                    want = true;
                                    tstart = md.x;
                                    tcount = (md.y - md.x) | ((op.flags & OP_LEAFBIT) ? kLeafAuthBit : 0u) | ((CMB && (op.flags & OP_ALL)) ? kAllBit : 0u);
                                    tmeta = make_meta(op.key, L + 1, key);
                                }
                            }
                        }
                    }
                    if (leafop) {
                        if (h) has[tcell] = 1;
                        else if (derr) err[tcell] = ITEM_ERR_DEPTH;
                    } else {
                        hit |= h;
                        depth_err |= derr;
                    }
                }
            }
            const uint64_t b = __ballot(want);
            if (b) {
                if (want) {
                    const uint32_t q = T + lanes_below(b);
                    t.a[q] = make_uint4(tstart, 0u, 1u, tcell);  // (the subject's row is looked up by the expansion that needs it)
                    t.count[q] = tcount;
                    t.meta[q] = tmeta;
                    t.sid[q] = sid;
                }
                T += (uint32_t)__popcll(b);
                seg_end[nseg++] = T;
                if (nseg == kMaxSeg || T > kTaskCap - 64) {  // list (nearly) full: expand what is recorded, then resume with the next op
                    jj++;
                    break;
                }
            }
        }
        jstart = jj;
        if (active) {
            if (hit) ans_set<E8 && ACL_ANS_LDS>(has, req, wo.first, 1);
            else if (depth_err) ans_set<E8 && ACL_ANS_LDS>(err, req, wo.first, ITEM_ERR_DEPTH);
        }
        const bool more = jstart < maxops;
        // ---- expand the recorded segments (the interpreter's state is dead from here to the loop's top)
        uint32_t a = 0;
#pragma unroll 1
        for (uint32_t k = 0; k < nseg; k++) {
            const uint32_t bnd = seg_end[k], cnt = bnd - a;  // cnt <= 64
            if (a) {  // bring the segment to the front of the list (expansions address tasks from 0)
                uint4 c0 = make_uint4(0, 0, 0, 0);
                uint32_t c1 = 0, c3 = 0, c4 = 0;
                if (lane < cnt) { c0 = t.a[a + lane]; c1 = t.count[a + lane]; c3 = t.meta[a + lane]; c4 = t.sid[a + lane]; }
                wave_lds_fence();
                if (lane < cnt) { t.a[lane] = c0; t.count[lane] = c1; t.meta[lane] = c3; t.sid[lane] = c4; }
            }
            flush_tasks<true, SHARDED, LOCAL, false, CMB>(t, cnt, wo, lane, g, progs, ops, g.edges, has, err, sh, false, co);
            a = bnd;
        }
        if (!more) break;
    }
    ACL_MARK(wo, PH_GENERIC);
}

// the program table (a few hundred bytes for real schemas) is copied into LDS when it fits
template <bool LDSPROG>
__device__ __forceinline__ void load_programs(const DevGraph &g, uint4 *s_prog, const SlotProg *&progs, const FwdOp *&ops, uint32_t nthreads) {
    progs = g.progs;
    ops = g.ops;
    if (LDSPROG) {
        const uint32_t np = g.nslots * 2, no = g.nops * 2;
        for (uint32_t i = threadIdx.x; i < np; i += nthreads) s_prog[i] = reinterpret_cast<const uint4 *>(g.progs)[i];
        for (uint32_t i = threadIdx.x; i < no; i += nthreads) s_prog[np + i] = reinterpret_cast<const uint4 *>(g.ops)[i];
        __syncthreads();
        progs = reinterpret_cast<const SlotProg *>(s_prog);
        ops = reinterpret_cast<const FwdOp *>(s_prog + np);
    }
}

__device__ __forceinline__ WaveOut chunked_out(const DevFrontier &f, uint32_t iter, uint32_t wave, WaveOutCold *cold, uint32_t lane) {
    if (lane == 0) {
        cold->counts = f.counts[iter & 1u];
        cold->nchunks = f.nchunks + iter;
        cold->overflow = f.overflow;
        cold->nwaves = f.nwaves;
        cold->max_chunks = f.max_chunks;
        cold->cap = 0;
    }
    wave_lds_fence();
    WaveOut wo;
    wo.buf = f.buf[iter & 1u];
    wo.cur = wave;
    wo.fill = 0;
    wo.produced = 0;
    wo.cold = cold;
    wo.lcap = 0;
    return wo;
}

// Segment-major walk over the chunks iteration `iter - 1` produced.  Chunks are mostly part-filled, so their low segments
// carry the work; walking all chunks' segment 0 first, then segment 1, ... spreads it evenly over the waves (each segment
// layer is rotated so that one wave does not keep landing on the same chunk).  The fill counts of the wave's next 64
// segment slots are fetched by its 64 lanes in ONE gather; the loop then reads them with readlane -- not one dependent
// load per slot (most slots are empty: 16+ per wave and level).
struct ChunkWalk {
    const uint4 *__restrict__ in;
    uint32_t lane;
    const uint32_t *slots;  // LDS, wave-private, of the wave's next 64 segment slots: [0, 64) first entry of the segment (chunk * kChunk + segment * 64),
                            // [64, 128) entries the chunk holds from that segment on -- in LDS, not in two VGPRs that live across every expansion
    uint64_t work;
    __device__ __forceinline__ void load(int wl, uint4 &e, bool &valid) const {
        const uint32_t at = uniform(slots[wl]), left = uniform(slots[64 + wl]);  // left >= 1: the slot's work bit is set
        valid = lane < left;
        e = gld(in, at + (valid ? lane : 0u));  // unconditional (a load under `if` is waited for where the branch rejoins); the caller masks by `valid`
    }
    __device__ __forceinline__ bool peek(uint4 &e, bool &valid) const {
        if (!work) return false;
        load(__ffsll((unsigned long long)work) - 1, e, valid);
        return true;
    }
    __device__ __forceinline__ void take() { work &= work - 1; }
};

template <bool LDSPROG, bool SHARDED, bool CMB = false>
__global__ __launch_bounds__(kBlock, ACL_MIN_WAVES_PER_SIMD) void k_expand(DevGraph g, DevFrontier f, uint32_t iter, uint8_t *has, uint8_t *err,
                                                                           DevShard sh) {
    __shared__ TaskLds lds[kWavesPerBlock];
    __shared__ WaveOutCold s_cold[kWavesPerBlock];
    __shared__ uint32_t s_slots[kWavesPerBlock][128];
    extern __shared__ uint4 s_prog[];  // dynamic: sized by the launcher to THIS snapshot's program table (a fixed 8 KiB cost two blocks per CU)
    const SlotProg *progs;
    const FwdOp *ops;
    load_programs<LDSPROG>(g, s_prog, progs, ops, kBlock);
    const uint32_t lane = lane_id();
    const uint32_t wib = uniform(threadIdx.x >> 6);  // (wave-uniform, and the compiler is told so: per-wave pointers then live in SGPRs)
    TaskLds &t = lds[wib];
    if (lane < kHeadWords) t.heads[lane] = 0ull;  // (flush_simple's head bits: zero between expansions)
    const uint32_t wave = blockIdx.x * kWavesPerBlock + wib, nwaves = f.nwaves;
    const uint32_t pin = (iter + 1) & 1u;  // iteration i reads parity (i-1)&1
    const uint32_t *__restrict__ in_counts = f.counts[pin];
    const bool live = !*f.overflow && f.any[iter - 1];
    const uint32_t C = live ? nwaves + min(f.nchunks[iter - 1], f.max_chunks - nwaves) : 0u;
    WaveOut wo = chunked_out(f, iter, wave, &s_cold[wib], lane);
    uint32_t *slots = s_slots[wib];
    ChunkWalk cw{f.buf[pin], lane, slots, 0ull};
    CombineOut co;
    if (CMB) co = CombineOut{g.nodes, g.ccount + 1, g.ccount, g.node_cap, g.cell_cap, g.cell0, iter, g.bexpr};
    const uint32_t nslot = C * kSegsPerChunk;
    for (uint32_t x0 = wave; x0 < nslot; x0 += 64 * nwaves) {
        // (once per 64 slots: keep the division's precomputed reciprocal out of a VGPR that would live across every expansion)
        uint32_t Cl = C;
        asm volatile("" : "+s"(Cl));
        const uint32_t xl = x0 + lane * nwaves;
        uint32_t lat = 0, lcnt = 0;
        if (xl < nslot) {
            const uint32_t ls = xl / Cl, lc = (xl % Cl + ls * 509u) % Cl;
            lcnt = in_counts[lc];
            lcnt = lcnt > ls * 64 ? lcnt - ls * 64 : 0u;  // entries of this slot's segment and beyond
            lat = lc * kChunk + ls * 64;
        }
        wave_lds_fence();  // (the previous round's readers are done)
        slots[lane] = lat;
        slots[64 + lane] = lcnt;
        cw.work = __ballot(lcnt != 0);
        wave_lds_fence();
        while (cw.work) {
            const int wl = __ffsll((unsigned long long)cw.work) - 1;
            cw.work &= cw.work - 1;
            uint4 e;
            bool valid;
            cw.load(wl, e, valid);
            process_segment<SHARDED, false, CMB>(e, valid, cw, t, wo, lane, g, progs, ops, has, err, sh, co);
        }
    }
    if (lane == 0) {
        if (wo.cur != kNoSpace) wo.cold->counts[wo.cur] = wo.fill;  // also publishes 0 for an unused static chunk
        if (wo.produced) f.any[iter] = 1u;
    }
}

// ------------------------------------------------------------ single launch
// Small batches -- the proxy's own call shape: one item per check expression (reference pkg/authz/check.go:76-94), one per
// watch update (watch.go:50), a few thousand after micro-batching.  The level-synchronous path costs a launch (and part of
// a host round trip) per dispatch level, ~17 us each whatever the batch size; here ONE launch answers the batch: every wave
// takes `rpw` requests, seeds them in registers, and walks them through all levels itself with a wave-private frontier
// (two regions of the frontier buffers, ping-pong), then writes the answers (k_seed + k_expand x levels + k_finalize fused).
// No wave ever reads another wave's entries, so there is no grid barrier and no inter-level visibility to arrange beyond
// the wave's own release/acquire.  A wave that outgrows its region raises `overflow`; the host redoes the batch on the
// level-synchronous path.  Same segment processor, same decisions.
template <bool E8>
struct LocalWalk {
    const uint4 *__restrict__ in;  // (E8: 8-byte entries, see put_entry)
    uint32_t n, s, lane;  // entries in the input region; the segment being processed (segments are claimed in pairs: s even)
    bool second;          // the pair's second segment is still to be taken
    const uint2 *sreq;    // E8: the unit's per-request constants {subject id, subject key} (LDS)
    uint32_t first;       // E8: the unit's first request
    uint2 rawB = make_uint2(0u, 0u);  // ACL_PREFETCH_ENTRIES: the pair's second segment, fetched (undecoded) when the pair was claimed
    bool have_rawB = false;
    __device__ __forceinline__ uint2 raw(uint32_t i) const { return gld(reinterpret_cast<const uint2 *>(in), i); }
    __device__ __forceinline__ uint4 at(uint32_t i) const {
        if (E8) return decode_entry8(raw(i), sreq, first);
        return gld(in, i);
    }
    __device__ __forceinline__ bool peek(uint4 &e, bool &valid) const {
        if (!second || (s + 1) * 64 >= n) return false;
        valid = (s + 1) * 64 + lane < n;
        if (E8 && have_rawB) e = decode_entry8(rawB, sreq, first);
        else e = at(valid ? (s + 1) * 64 + lane : (s + 1) * 64);  // unconditional, like ChunkWalk::load
        return true;
    }
    __device__ __forceinline__ void take() { second = false; }
};
struct NoNext {
    __device__ __forceinline__ bool peek(uint4 &, bool &) const { return false; }
    __device__ __forceinline__ void take() {}
};

// Work = units of `rpw` (<= 256) consecutive requests, one unit per BLOCK at a time (block b takes units b, b + nblocks, ...).
// The block's four waves walk the unit together, level by level, inside the block's private frontier region: a level's
// segments are claimed pair by pair through an LDS counter, children are appended through an LDS cursor, and one block
// barrier separates the levels.  Sharing a unit between four waves is what keeps the launch's tail short: requests differ
// 100-fold in work (a first-level hit against a full five-level miss), and a wave that walked its own 43 requests alone
// finished anywhere between 0.5x and 1.6x the mean -- a third of the wave-time was idle waiting for the slowest
// (profiles/r02_pmc_walk.txt); a unit of 4 x 43 requests varies half as much, and inside it the work is shared per level.
// Block b starts on unit b; further units (batches beyond 256 requests per resident block, or `upw` > 1) are handed out through
// `next_unit`, one atomic per BLOCK-unit -- per wave-unit the same counter cost ~12 ns per unit in same-address contention
// (profiles/r02_walk_units_per_wave.txt).
struct InlineItems {  // up to four 16-byte items passed by value (kernel arguments)
    uint4 v[4];
};
#ifndef ACL_LOCAL_WAVES_PER_SIMD
#define ACL_LOCAL_WAVES_PER_SIMD 6  // (round 5; 8 in rounds 2-4: see ACL_LOCAL_WIDE)
#endif
// Waves per block = waves that share one unit.  Two instantiations: kLocalNarrow for batches that do not fill the chip (a unit is a handful of
// requests: more, smaller blocks) and kLocalWide for chip-filling ones -- requests differ 100-fold in work, so the more requests (and waves) a
// unit pools, the less the slowest block's sum sticks out: C4's 262 144-item batch 303 us with 4 waves per block (2 048 units of 128 requests),
// 286 us with 8, 276 us with 16 (512 units of 512), same-box A/B in profiles/r03_waves_per_block_ab.txt.
#ifndef ACL_LOCAL_WIDE
#define ACL_LOCAL_WIDE 12  // Round 5: 12 waves per block, two blocks per CU = 6 waves per SIMD with 72 VGPRs and three children per lane and step -- C4 221.4 -> 216.8 us, the
                           // 100 M-relationship replica 294.4 -> 283.8 us (16 waves at 64 VGPRs and two children per step until then: every attempt to add live state to
                           // the fast paths at 64 VGPRs ended in spills; 12 waves had lost to 16 in round 4, when a child still cost two bucket gathers = 8 VGPRs more).
                           // 14 waves: 275 us, 10 waves x 3 blocks: 226 us, four children per step: 227 us (profiles/r05_ab_block_shape.txt)
#endif
constexpr int kLocalNarrow = 4, kLocalWide = ACL_LOCAL_WIDE;
#ifndef ACL_SPLIT_UNITS
#define ACL_SPLIT_UNITS 0  // 1 (A/B builds; measured C4 228.4 -> 226.0 us but C5R 297 -> 304 us, level barriers 19 -> 15 % of the wave-time: profiles/r05_ab_split_units.txt): the wide monotone walk cuts a unit into two HALVES whose levels turn over independently (k_check_local); 0 = one barrier per level (A/B builds)
#endif
#ifndef ACL_PREFETCH_ENTRIES
#define ACL_PREFETCH_ENTRIES 1  // the single-launch walk fetches the NEXT pair's entries before it expands the current pair (k_check_local's claim loop); 0 = A/B builds
#endif
#ifndef ACL_TAIL_SINGLES
#define ACL_TAIL_SINGLES 0  // N > 0 (A/B builds): the last 2 N x WAVES segments of a level are claimed one by one instead of in pairs (see the claim loop)
#endif
template <bool LDSPROG, int WAVES, bool CMB = false>
__global__ __launch_bounds__(WAVES * 64, ACL_LOCAL_WAVES_PER_SIMD) void k_check_local(DevGraph g, const uint4 *__restrict__ items, uint32_t n, uint32_t rpw,
                                                                                  uint32_t nunits, uint32_t nstatic, uint32_t rdyn, uint32_t *next_unit, uint4 *buf0, uint4 *buf1,
                                                                                  uint32_t cap,
                                                                                  uint32_t *overflow, uint8_t *has, uint8_t *err, uint8_t *perm_out,
                                                                                  int32_t *err_out, uint32_t *max_level, uint32_t skew, uint32_t *done_ctr, uint32_t *done_flag,
                                                                                  uint32_t done_val, InlineItems inl) {
    __shared__ TaskLds lds[WAVES];
    __shared__ WaveOutCold s_cold[WAVES];
    // output cursor / segment-claim counter of level L live in slot L % 3: written during L, read at the start of L + 1, cleared at the
    // start of L + 2 (every wave has read them by then) and reused at L + 3 -- ONE block barrier per level instead of three
    // NH = 2 (chip-filling monotone batches): the unit's requests are cut into two halves with cursor sets, frontier regions and arrival counters of
    // their own, walked by the same waves in alternation -- see the phase loop below
    constexpr uint32_t NH = (ACL_SPLIT_UNITS && !CMB && WAVES >= 8) ? 2u : 1u;
    __shared__ uint32_t s_cursors[6 * NH + NH], s_stop, s_unit;  // (one array: the per-unit reset is ONE store through one address -- two arrays cost a spilled VGPR)
    uint32_t *const s_fill = s_cursors, *const s_next = s_cursors + 3;  // half h: s_fill + 6 h, s_next + 6 h; s_done = s_cursors + 6 NH
    uint32_t *const s_done = s_cursors + 6 * NH;
    __shared__ uint32_t s_ccount[2];  // CMB: {leaf cells, nodes} of the unit being walked
    constexpr bool E8 = ACL_ENTRY8 && !CMB;  // 8-byte frontier entries (put_entry)
    __shared__ uint2 s_req[E8 ? WAVES * 64 : 1];  // E8: {subject id, subject key} of the unit's requests
    constexpr bool AL = E8 && ACL_ANS_LDS;
    __shared__ uint8_t s_has[AL ? WAVES * 64 : 1], s_err[AL ? WAVES * 64 : 1];  // the unit's answer bytes (ans_get / ans_set)
    if (AL) {
        has = s_has;
        err = s_err;
    }
    extern __shared__ uint4 s_prog[];  // dynamic: sized by the launcher to THIS snapshot's program table (a fixed 8 KiB cost two blocks per CU)
    const SlotProg *progs;
    const FwdOp *ops;
    load_programs<LDSPROG>(g, s_prog, progs, ops, WAVES * 64);
    CombineOut co;  // the block's own region of nodes and leaf cells
    if (CMB) co = CombineOut{g.nodes + (size_t)blockIdx.x * g.node_cap, &s_ccount[1], &s_ccount[0], g.node_cap, g.cell_cap, g.cell0 + blockIdx.x * g.cell_cap, 1u, g.bexpr};
    const uint32_t lane = lane_id();
    const uint32_t wib = uniform(threadIdx.x >> 6);  // (wave-uniform, and the compiler is told so: per-wave pointers then live in SGPRs)
    TaskLds &t = lds[wib];
    if (lane < kHeadWords) t.heads[lane] = 0ull;  // (flush_simple's head bits: zero between expansions)
    const DevShard nosh{};
    uint4 *bufs[2] = {buf0 + (size_t)blockIdx.x * cap, buf1 + (size_t)blockIdx.x * cap};
    if (lane == 0) {
        s_cold[wib] = WaveOutCold{};
        s_cold[wib].overflow = overflow;
        s_cold[wib].cap = E8 ? 2u * cap : cap;  // (the block's region holds twice as many 8-byte entries)
#if ACL_PROFILE_PHASES
        s_cold[wib].last = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    }
    if (threadIdx.x == 0) s_stop = 0;
    WaveOut wo;
    wo.cur = 0;
    wo.fill = 0;
    wo.produced = 0;
    wo.cold = &s_cold[wib];
    wo.lcap = E8 ? 2u * cap : cap;  // (the block's region holds twice as many 8-byte entries)
    wo.lfill = &s_fill[1];
    // units [0, nstatic) hold rpw requests each (block b starts on unit b: no hand-out); the requests behind them come in SMALL units of rdyn,
    // handed out through `next_unit` as blocks finish -- the launch's tail is then a small unit's walk, not the slowest big unit's
    const uint32_t nstat_req = min(n, nstatic * rpw);
    for (uint32_t unit = blockIdx.x; unit < nunits;) {
        // skew != 0 (host-mapped batches of a lone caller, static units only): the items cross PCIe in block order, so block u's seeds arrive
        // ~ u / nunits of the transfer late -- unit u gets rpw + skew (first block) ... rpw - skew (last) requests, and the blocks end together
        // instead of the last ones trailing by the transfer time: boundary(u) = u rpw + skew u (nunits - u) / nunits, `skew` in 1/256ths per unit
        uint32_t first = unit < nstatic ? unit * rpw : nstat_req + (unit - nstatic) * rdyn;
        uint32_t mine = unit < nstatic ? min(rpw, nstat_req - first) : min(rdyn, n - first);  // <= WAVES * 64: thread i seeds and answers request first + i
        if (skew) {
            const uint32_t u = uniform(unit);
            first = min(n, u * rpw + ((skew * u * (nunits - u)) >> 8));
            mine = min(n, (u + 1u) * rpw + ((skew * (u + 1u) * (nunits - u - 1u)) >> 8)) - first;
        }
        if (threadIdx.x < 6 * NH + NH) s_cursors[threadIdx.x] = 0;
        if (CMB && threadIdx.x < 2) s_ccount[threadIdx.x] = 0;
        __syncthreads();
        // ---- seeds (k_seed's validation), in registers: wave w holds requests [64 w, 64 w + 64) of the unit
        const bool valid = threadIdx.x < mine;
        const uint32_t req = first + threadIdx.x;
        uint4 e = make_uint4(0, 0, kDeadMeta, 0);
        if (valid) {
            // (items == nullptr: a batch of <= 4 items rides in the kernel's arguments -- a single check's item does not cost a trip across PCIe)
            uint4 it;
            if (items) {  // (uniform)
                it = gld(items, req);
            } else {
                const uint32_t qi = req & 3u;
                it = qi == 0u ? inl.v[0] : (qi == 1u ? inl.v[1] : (qi == 2u ? inl.v[2] : inl.v[3]));
            }
            const uint32_t rtype = it.x & 0xFFFFu, perm = it.x >> 16, stype = it.z & 0xFFFFu, srel = it.z >> 16;
            const bool tok = rtype < g.ntypes && stype < g.ntypes;
            const uint32_t rt = tok ? rtype : 0u, st = tok ? stype : 0u;
            const uint32_t rmem = gld(g.type_nmembers, rt), smem = gld(g.type_nmembers, st), rbase = gld(g.type_slot_base, rt), sbase = gld(g.type_slot_base, st);
            const bool ok = tok && perm < rmem && (srel == 0xFFFFu || srel < smem);
            ans_set<E8 && ACL_ANS_LDS>(has, req, first, 0);
            ans_set<E8 && ACL_ANS_LDS>(err, req, first, (uint8_t)(ok ? ITEM_ERR_NONE : ITEM_ERR_INVALID));
            const uint32_t skey = srel == 0xFFFFu ? g.nslots + stype : sbase + srel;
            const uint32_t meta = ok ? make_meta(rbase + perm, 1u, skey) : kDeadMeta;
            e = make_uint4(it.y, req, meta, it.w);
            if (E8) s_req[threadIdx.x] = make_uint2(it.w, skey);
        }
        wo.first = first;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t level_reached = 1;
        if (NH == 2) {
            // ---- TWO HALF-UNITS, no block barrier between levels (round 5; VERDICT r4 next #1c).  With one cursor set a level ends at a block
            // barrier: a wave that finds the claim counter dry waits there for the waves still expanding their last pair -- a fifth of the walk's
            // wave-time (profiles/r04_phases_final.txt), because a unit's level is only a few dozen segment pairs for a block's waves.  Here the unit's
            // requests are cut in two (at a multiple of 64: a wave's seeds belong to one half), each half with its own cursors, frontier region
            // (half of the block's) and an ARRIVAL COUNTER; every wave walks the phases  (half 0, level 2), (half 1, level 2), (half 0, level 3), ...
            // in that order and, done with its share of a phase, goes straight on to the next one -- which only needs the OTHER half's previous
            // level complete, long finished as a rule.  A phase is complete when all WAVES waves have arrived at its counter (monotonic: phase
            // (h, L) needs s_done[h] >= WAVES (L - 2)); the counter is raised behind a workgroup release fence and read before an acquire fence,
            // the pairing __syncthreads() provided.  Cursor slot (L + 1) % 3 of a half is cleared by every wave that enters (h, L): nobody uses
            // it between the end of (h, L - 1) -- all waves have arrived there -- and the start of (h, L + 1).
            const uint32_t split = min(mine, ((mine / 2u + 63u) >> 6) << 6);  // requests [0, split) of the unit are half 0
            const uint32_t hcap = wo.cold->cap >> 1;                          // entries of a half's region (the same in both buffers)
            const uint32_t myh = (wib * 64u >= split) ? 1u : 0u;              // the half this wave's seeds belong to
            const size_t hoff = (size_t)(E8 ? hcap / 2u : hcap);              // ... in uint4 units
            if (lane == 0) wo.cold->cap = hcap;
            wo.lcap = hcap;
            wo.buf = bufs[0] + myh * hoff;
            wo.cur = 0;
            wo.lfill = &s_fill[6 * myh + 1];  // level 1
            ACL_MARK(wo, PH_SEED);
            {
                NoNext nn;
                co.iter = 1u;
                process_segment<false, true, CMB, NoNext, true>(e, valid, nn, t, wo, lane, g, progs, ops, has, err, nosh, co);
            }
            if (wo.cur == kNoSpace && lane == 0) s_stop = 1;
            __syncthreads();  // (the seeds' children are written; from here on the halves' counters order everything)
            uint32_t lvl[2] = {2u, 2u}, par[2] = {0u, 0u};
            bool alive[2] = {true, true};
            for (uint32_t ph = 0; alive[0] || alive[1]; ph++) {
                const uint32_t h = ph & 1u;
                if (!alive[h]) continue;
                const uint32_t level = lvl[h];
                if (level > kMaxLevels + 1) {
                    alive[h] = false;
                    continue;
                }
                // ---- every wave has left (h, level - 1): its entries are written, its cursors final
                const uint32_t target = WAVES * (level - 2u);
                while (__hip_atomic_load(&s_done[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && !__hip_atomic_load(&s_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                ACL_MARK(wo, PH_BARRIER);
                if (__hip_atomic_load(&s_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;  // overflow somewhere: the host redoes the batch
                uint32_t *const fill = s_fill + 6 * h, *const nexts = s_next + 6 * h;
                const uint32_t cnt = uniform(fill[(level - 1) % 3]);
                if (lane == 0) {
                    fill[(level + 1) % 3] = 0;
                    nexts[(level + 1) % 3] = 0;
                }
                if (!cnt) {
                    alive[h] = false;
                    continue;
                }
                level_reached = max(level_reached, level);
                wo.lfill = &fill[level % 3];
                uint32_t *const next_seg = &nexts[level % 3];
                LocalWalk<E8> lw{bufs[par[h]] + h * hoff, cnt, 0u, lane, false, s_req, first};
                par[h] ^= 1u;
                wo.buf = bufs[par[h]] + h * hoff;
                co.iter = level;
                const uint32_t nseg = (cnt + 63u) >> 6;
                for (;;) {
                    uint32_t sg = 0;
                    if (lane == 0) sg = atomicAdd(next_seg, 2u);
                    sg = uniform(sg);
                    if (sg >= nseg) break;
                    for (lw.s = sg, lw.second = true; lw.s < sg + 2 && lw.s * 64 < cnt; lw.s++) {
                        if (lw.s > sg && !lw.second) break;  // the pair's second segment went with the first
                        const bool v = lw.s * 64 + lane < cnt;
                        const uint4 en = lw.at(v ? lw.s * 64 + lane : lw.s * 64);  // unconditional; process_segment masks by `v`
                        if (lw.s > sg) lw.second = false;
                        process_segment<false, true, CMB>(en, v, lw, t, wo, lane, g, progs, ops, has, err, nosh, co);
                    }
                }
                if (wo.cur == kNoSpace && lane == 0) s_stop = 1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's entries and answer bytes before its arrival
                if (lane == 0) __hip_atomic_fetch_add(&s_done[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lvl[h] = level + 1u;
            }
            if (lane == 0) wo.cold->cap = hcap << 1;  // (the next unit sets its own geometry from the full region)
            wo.lcap = hcap << 1;
        } else {
        wo.buf = bufs[0];
        wo.cur = 0;
        wo.lfill = &s_fill[1];  // level 1
        ACL_MARK(wo, PH_SEED);
        {
            NoNext nn;
            co.iter = 1u;
            process_segment<false, true, CMB, NoNext, true>(e, valid, nn, t, wo, lane, g, progs, ops, has, err, nosh, co);
        }
        uint32_t parity = 0;
        for (uint32_t level = 2; level <= kMaxLevels + 1; level++) {
            // ---- level boundary: everybody's children are written, the cursors turn over
            if (wo.cur == kNoSpace && lane == 0) s_stop = 1;  // overflow: the host redoes the batch
            __syncthreads();
            const uint32_t cnt = s_fill[(level - 1) % 3];
            const bool stop = s_stop != 0;
            if (threadIdx.x == 0) {
                s_fill[(level + 1) % 3] = 0;
                s_next[(level + 1) % 3] = 0;
            }
            ACL_MARK(wo, PH_BARRIER);
            if (stop || !cnt) break;
            level_reached = level;
            wo.lfill = &s_fill[level % 3];
            uint32_t *const next_seg = &s_next[level % 3];
            LocalWalk<E8> lw{bufs[parity], cnt, 0u, lane, false, s_req, first};
            parity ^= 1u;
            wo.buf = bufs[parity];
            co.iter = level;
            // Segments are claimed in PAIRS: one prologue and one set of trips for 128 entries.  (ACL_TAIL_SINGLES = N > 0, A/B builds: the level's last
            // 2 N x WAVES segments one by one, so that what a wave still holds when the counter runs dry is half as long -- the level barriers are a
            // fifth of the walk's wave-time.  Measured slower, 227.5 -> 231 -> 238 us for N = 0, 1, 2 on C4: profiles/r05_ab_seeded_rows.txt.)
            const uint32_t nseg = (cnt + 63u) >> 6;
            const uint32_t npair = (ACL_TAIL_SINGLES && nseg > 2u * WAVES * ACL_TAIL_SINGLES) ? (nseg - 2u * WAVES * ACL_TAIL_SINGLES + 1u) >> 1 : (ACL_TAIL_SINGLES ? 0u : nseg);
#if ACL_PREFETCH_ENTRIES
            if (E8 && !ACL_TAIL_SINGLES) {
                // ---- the NEXT pair's entries travel while the current pair is expanded (round 6).  A pair's walk is a chain of dependent trips -- its entries,
                // the parents' descriptors, then edges and buckets per step -- and the first of them, the entries (an L2 hit: this block wrote them a level
                // ago), was a tenth of the walk's wave-time (profiles/r04_phases_final.txt "entries wait").  A wave now claims pair k + 1 and issues the
                // loads of its 128 raw 8-byte entries (4 VGPRs) BEFORE it expands pair k; they are decoded when their turn comes.  Claiming ahead
                // is only done while at least a round of pairs is still unclaimed: a pair held by a busy wave while other waves stand at the level
                // barrier would lengthen the level's tail (the last WAVES pairs are claimed the old way, when their wave is free).
                auto claim = [&]() -> uint32_t {
                    uint32_t c = 0;
                    if (lane == 0) c = atomicAdd(next_seg, 1u);
                    return 2u * uniform(c);
                };
                auto fetch = [&](uint32_t sgp, uint2 &ra, uint2 &rb) {
                    const uint32_t last = cnt - 1u;
                    ra = lw.raw(min(sgp * 64u + lane, last));
                    rb = lw.raw(min(sgp * 64u + 64u + lane, last));
                };
                uint32_t sg = claim();
                uint2 rA = make_uint2(0u, 0u), rB = rA;
                if (sg < nseg) fetch(sg, rA, rB);
                while (sg < nseg) {
                    uint32_t sgn = nseg;
                    uint2 nA = make_uint2(0u, 0u), nB = nA;
                    const uint32_t seen = uniform(__hip_atomic_load(next_seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const bool ahead = 2u * (seen + (uint32_t)WAVES) < nseg;
                    if (ahead) {
                        sgn = claim();
                        if (sgn < nseg) fetch(sgn, nA, nB);
                    }
                    {
                        lw.s = sg;
                        lw.second = true;
                        lw.rawB = rB;
                        lw.have_rawB = true;
                        const bool v = sg * 64u + lane < cnt;
                        process_segment<false, true, CMB>(decode_entry8(rA, s_req, first), v, lw, t, wo, lane, g, progs, ops, has, err, nosh, co);
                        if (lw.second && (sg + 1u) * 64u < cnt) {  // the pair's second segment did not go with the first
                            lw.s = sg + 1u;
                            lw.second = false;
                            const bool vb = (sg + 1u) * 64u + lane < cnt;
                            process_segment<false, true, CMB>(decode_entry8(rB, s_req, first), vb, lw, t, wo, lane, g, progs, ops, has, err, nosh, co);
                        }
                    }
                    if (!ahead) {
                        sgn = claim();
                        if (sgn < nseg) fetch(sgn, nA, nB);
                    }
                    sg = sgn;
                    rA = nA;
                    rB = nB;
                }
            } else
#endif
            for (;;) {
                uint32_t sg = 0;
                if (lane == 0) sg = atomicAdd(next_seg, 1u);
                sg = uniform(sg);
                const uint32_t take = sg < npair ? 2u : 1u;
                sg = sg < npair ? 2u * sg : npair + sg;  // (= 2 npair + (claim - npair))
                if (sg >= nseg) break;
                for (lw.s = sg, lw.second = take > 1u; lw.s < sg + take && lw.s * 64 < cnt; lw.s++) {
                    if (lw.s > sg && !lw.second) break;  // the pair's second segment went with the first
                    const bool v = lw.s * 64 + lane < cnt;
                    const uint4 en = lw.at(v ? lw.s * 64 + lane : lw.s * 64);  // unconditional; process_segment masks by `v`
                    if (lw.s > sg) lw.second = false;
                    process_segment<false, true, CMB>(en, v, lw, t, wo, lane, g, progs, ops, has, err, nosh, co);
                }
            }
        }
        }
        // (statistics: dispatch levels the deepest request of the batch needed.  Test before the atomic: 2 048 blocks ending together on
        //  one address serialise at ~12 ns each -- C2's 18 us kernel took 38 us with an unconditional atomicMax.)
        if (max_level && threadIdx.x == 0 && level_reached > __hip_atomic_load(max_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_level, level_reached);
        // ---- combine nodes: deepest iteration first (a node's leaves are answered by the walk and by nodes of LATER iterations only)
        if (CMB) {
            __syncthreads();
            const uint32_t nn = min(s_ccount[1], co.node_cap);
            if (nn && !s_stop) {
                for (uint32_t it = level_reached; it >= 1u; it--) {
                    for (uint32_t ph = 0; ph < 2u; ph++) {  // the member nodes of intersection arrows first: the regular nodes of `it` read what they fold
                        for (uint32_t i = threadIdx.x; i < nn; i += WAVES * 64) {
                            const uint4 nd = co.nodes[i];
                            if ((nd.z >> 16) != it || (nd.w != 0u) != (ph == 0u)) continue;
                            if (ph == 0u) resolve_member(nd, has, err);
                            else resolve_node(nd, progs, co.bexpr, has, err);
                        }
                        __syncthreads();
                    }
                }
            }
        }
        // ---- answers (k_finalize): every wave's has[] / err[] stores are behind a block barrier
        __syncthreads();
        if (valid) {
            const bool h = AL ? s_has[threadIdx.x] != 0 : __hip_atomic_load(has + req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            const uint8_t er = h ? (uint8_t)ITEM_ERR_NONE : (AL ? s_err[threadIdx.x] : __hip_atomic_load(err + req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            perm_out[req] = h ? 2 : (er ? 0 : 1);
            if (err_out) err_out[req] = er == ITEM_ERR_DEPTH ? 100 : (er == ITEM_ERR_INVALID ? 9 : 0);
        }
        if (s_stop || !next_unit) break;
        if (threadIdx.x == 0) s_unit = gridDim.x + atomicAdd(next_unit, 1u);
        __syncthreads();
        unit = s_unit;
    }
#if ACL_PROFILE_PHASES
    ACL_MARK(wo, PH_OTHER);
    if (lane < 16) atomicAdd(&acl_phase_cycles[lane], (unsigned long long)s_cold[wib].prof[lane]);
#endif
    // ---- small host batches (round 5): the caller does not wait in hipStreamSynchronize -- that costs ~5.5 us AFTER the kernel's
    // last store is visible to a spinning host thread (tools/launch_latency.hip: flag visible 6.6 us after the launch call, synchronize returns at
    // 11.9-15 us) -- it spins on `done_flag`, a word of pinned host memory that the LAST block to finish sets.  Every block releases its answers
    // (stores into host memory) at system scope before it arrives at the device counter; the last arrival re-arms the counter and raises the flag.
    if (done_flag) {
        __syncthreads();  // (every wave's answer stores are issued)
        if (threadIdx.x == 0) {
            __threadfence_system();
            if (__hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
                __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done_flag, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Merges identical pending sub-checks of one level: two frontier entries with the same (request, state, level) have identical
// subtrees, so all but one are struck (meta = dead) -- answers cannot change.  Only run when a pass outgrew its frontier: on
// sane graphs duplicates are rare (1.8 % on C4, profiles/r02_c4_duplicate_ratio.txt) and the walk is faster without it, but group
// nesting with BRANCHING cycles (g0 -> {g0, g1}, g1 -> g0) doubles the frontier every level for 50 levels; SpiceDB's
// dispatcher cuts the same recursion at depth 50, and the oracle memoises on exactly this key.
// key = request[14] | level[6] | slot[13] | object id[31] (dedup passes run on slices of <= 16 384 requests); table: open
// addressing over u64, empty = ~0.
__global__ __launch_bounds__(256) void k_dedup(DevFrontier f, uint32_t iter, unsigned long long *table, uint32_t bits) {
    uint4 *buf = f.buf[iter & 1u];
    const uint32_t *counts = f.counts[iter & 1u];
    if (*f.overflow) return;
    const uint32_t C = f.nwaves + min(f.nchunks[iter], f.max_chunks - f.nwaves);
    const uint32_t mask = (1u << bits) - 1u;
    for (uint32_t c = blockIdx.x; c < C; c += gridDim.x) {
        const uint32_t cnt = counts[c];
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const uint4 e = buf[(size_t)c * kChunk + i];
            if (e.z == kDeadMeta) continue;
            const unsigned long long key = ((unsigned long long)(e.y & 0x3FFFu) << 50) | ((unsigned long long)meta_level(e.z) << 44) |
                                           ((unsigned long long)meta_slot(e.z) << 31) | (unsigned long long)(e.x & kIdMask);
            uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - bits));
            for (;;) {
                const unsigned long long old = atomicCAS(table + h, ~0ull, key);
                if (old == ~0ull) break;  // first of its kind
                if (old == key) {
                    buf[(size_t)c * kChunk + i].z = kDeadMeta;
                    break;
                }
                h = (h + 1u) & mask;
            }
        }
    }
}

// The same for schemas with `&` / `-`, whose entries carry a result CELL (32 bits) where the request was: cell | object id is the 64-bit key
// of the table, level | slot sits in a second table the slot's winner fills right behind its CAS -- a lane that meets its own first half
// reads the second one, and comes back in the next round while it is still empty (the winner has left the loop by then: no lane ever waits
// for a lane of its own wave).  Identical (cell, state, level) entries have identical subtrees AND identical combine nodes to create, so
// all but one are struck before any node exists.
__global__ __launch_bounds__(256) void k_dedup_cells(DevFrontier f, uint32_t iter, unsigned long long *table, uint32_t *second, uint32_t bits) {
    uint4 *buf = f.buf[iter & 1u];
    const uint32_t *counts = f.counts[iter & 1u];
    if (*f.overflow) return;
    const uint32_t C = f.nwaves + min(f.nchunks[iter], f.max_chunks - f.nwaves);
    const uint32_t mask = (1u << bits) - 1u;
    for (uint32_t c = blockIdx.x; c < C; c += gridDim.x) {
        const uint32_t cnt = counts[c];
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const uint4 e = buf[(size_t)c * kChunk + i];
            if (e.z == kDeadMeta) continue;
            const unsigned long long key = ((unsigned long long)e.y << 32) | (unsigned long long)(e.x & kIdMask);  // (bit 31 clear: never the empty mark)
            const uint32_t rest = 1u + (e.z & 0x7FFFFu);                                                           // level | slot, never 0
            uint32_t h = (uint32_t)(((key ^ ((unsigned long long)rest << 40)) * 0x9E3779B97F4A7C15ull) >> (64 - bits));
            for (;;) {
                const unsigned long long old = atomicCAS(table + h, ~0ull, key);
                if (old == ~0ull) {  // first of its kind: publish the second half
                    __hip_atomic_store(second + h, rest, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (old == key) {
                    const uint32_t r2 = __hip_atomic_load(second + h, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (r2 == 0u) continue;  // the winner is between its CAS and its store: look again
                    if (r2 == rest) {
                        buf[(size_t)c * kChunk + i].z = kDeadMeta;
                        break;
                    }
                }
                h = (h + 1u) & mask;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_finalize(uint32_t n, const uint8_t *__restrict__ has, const uint8_t *__restrict__ err, uint8_t *perm_out,
                                                   int32_t *err_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool h = has[i];
    const uint8_t e = h ? (uint8_t)ITEM_ERR_NONE : err[i];
    perm_out[i] = h ? 2 : (e ? 0 : 1);
    if (err_out) err_out[i] = e == ITEM_ERR_DEPTH ? 100 : (e == ITEM_ERR_INVALID ? 9 : 0);
}

// Level loop, schemas with `&` / `-`: the combine nodes of ONE frontier iteration (launched for iter = last .. 1; the list is scanned whole
// every time -- this is the overflow path, not the fast one).
__global__ __launch_bounds__(256) void k_resolve(DevGraph g, uint32_t iter, uint32_t members, uint8_t *has, uint8_t *err) {
    const uint32_t nn = min(g.ccount[1], g.node_cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        const uint4 nd = g.nodes[i];
        if ((nd.z >> 16) != iter || (nd.w != 0u) != (members != 0u)) continue;
        if (members) resolve_member(nd, has, err);  // (launched first: the regular nodes of the iteration read what the members fold)
        else resolve_node(nd, g.progs, g.bexpr, has, err);
    }
}

// Sharded graph, schemas with `&` / `-` (round 5): every shard appended the combine nodes of the states IT visited; after the walk the node lists are
// all-gathered (world blocks of `stride` nodes, block b holding hdrs[b].x of them) and, has / err being identical on every shard by then (a
// byte-wise max over the whole cell space), every shard resolves ALL nodes -- redundantly and identically, so no cell has to travel between
// the resolve iterations.
__global__ __launch_bounds__(256) void k_resolve_gathered(DevGraph g, const uint4 *__restrict__ nodes, uint32_t stride, const uint4 *__restrict__ hdrs, uint32_t iter, uint32_t members,
                                                          uint8_t *has, uint8_t *err) {
    const uint32_t nn = min(hdrs[blockIdx.y].x, stride);
    const uint4 *__restrict__ mine = nodes + (size_t)blockIdx.y * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        const uint4 nd = mine[i];
        if ((nd.z >> 16) != iter || (nd.w != 0u) != (members != 0u)) continue;
        if (members) resolve_member(nd, has, err);
        else resolve_node(nd, g.progs, g.bexpr, has, err);
    }
}
__global__ void k_node_hdr(uint4 *hdr, const uint32_t *ccount) { *hdr = make_uint4(ccount[1], ccount[0], 0u, 0u); }  // {nodes appended, cells handed out}

// Post-filter hand-off (reference pkg/authz/postfilter.go:144-178): list item i owns the bulk-check pairs
// [item_off[i], item_off[i+1]); it is kept iff every one of them is HAS_PERMISSION without error -- an item
// with no pairs (templates that did not resolve, postfilter.go:92-95,145-150) is kept.
__global__ __launch_bounds__(256) void k_keep(uint32_t k_items, const uint32_t *__restrict__ item_off, const uint8_t *__restrict__ perm, uint8_t *keep_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k_items) return;
    bool all = true;
    for (uint32_t j = item_off[i], e = item_off[i + 1]; j < e; j++) all = all && perm[j] == 2;  // PERMISSIONSHIP_HAS_PERMISSION
    keep_out[i] = all ? 1 : 0;
}

// ----------------------------------------------------------- reverse expand
// entry: x = object id, y = lookup request, z = meta (slot | dist << 13 | flags), w unused.
// dist == 0 marks a seed entry: slot field holds the SUBJECT KEY and the program is rseeds[key].
// PHASE: REV_FUSED  (unsharded) visit + expand in one pass;
//        REV_VISIT  (sharded)   test-and-set `visited`, pass first visits on (kRevVisited) and export the ones whose
//                               parent rows also live on other shards (kRevForeign) -- no expansion;
//        REV_EXPAND (sharded)   expand seeds / visited / foreign entries, never touching `visited`.
// Two phases keep the sharded walk level-synchronous: a state and its foreign copies are expanded in the same
// iteration, so "first visit wins" still records the minimum distance (the depth-50 cut depends on it).
template <uint32_t PHASE>
__global__ __launch_bounds__(kBlock, ACL_MIN_WAVES_PER_SIMD) void k_rev_expand(DevReverse r, DevFrontier f, uint32_t iter, DevShard sh) {
    __shared__ TaskLds lds[kWavesPerBlock];
    __shared__ WaveOutCold s_cold[kWavesPerBlock];
    const uint32_t lane = lane_id();
    const uint32_t wib = uniform(threadIdx.x >> 6);  // (wave-uniform, and the compiler is told so: per-wave pointers then live in SGPRs)
    TaskLds &t = lds[wib];
    const uint32_t wave = blockIdx.x * kWavesPerBlock + wib, nwaves = f.nwaves;
    const uint32_t pin = (iter + 1) & 1u;
    const uint4 *__restrict__ in = f.buf[pin];
    const uint32_t *__restrict__ in_counts = f.counts[pin];
    const bool live = !*f.overflow && f.any[iter - 1];
    const uint32_t C = live ? nwaves + min(f.nchunks[iter - 1], f.max_chunks - nwaves) : 0u;
    const DevGraph nog{};
    WaveOut wo = chunked_out(f, iter, wave, &s_cold[wib], lane);
    uint4 *__restrict__ out = wo.buf;
    // segment-major work order, as in k_expand (ChunkWalk)
    const uint32_t nslot = C * kSegsPerChunk;
    for (uint32_t x0 = wave; x0 < nslot; x0 += 64 * nwaves) {
        const uint32_t xl = x0 + lane * nwaves;
        uint32_t lc = 0, lcnt = 0;
        if (xl < nslot) {
            const uint32_t ls = xl / C;
            lc = (xl % C + ls * 509u) % C;
            lcnt = in_counts[lc];
            lcnt = lcnt > ls * 64 ? lcnt - ls * 64 : 0u;  // entries of this slot's segment and beyond
        }
        uint64_t work = __ballot(lcnt != 0);
        while (work) {
        const int wl = __ffsll((unsigned long long)work) - 1;
        work &= work - 1;
        const uint32_t x = x0 + (uint32_t)wl * nwaves;
        const uint32_t s = x / C, c = (uint32_t)__builtin_amdgcn_readlane((int)lc, wl);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)lcnt, wl) + s * 64;
        const bool valid = s * 64 + lane < cnt;
        const uint4 e = valid ? gld(in, c * kChunk + s * 64 + lane) : make_uint4(0, 0, kDeadMeta, 0);
        const uint32_t id = e.x, req = e.y, meta = e.z;
        bool active = valid && meta != kDeadMeta;
        const uint32_t slot = meta & 0x1FFFu, dist = (meta >> 13) & 63u;
        RevProg p{0, 0};
        bool first_visit = false;
        if (active) {
            if (dist == 0) {
                p = r.rseeds[slot];
            } else if (PHASE == REV_EXPAND) {
                p = r.rprogs[slot];  // kRevVisited or kRevForeign: the visit already happened (here or on the owner)
            } else if (id < r.slot_nobjects[slot]) {
                // first visit wins: level-synchronous order makes it the minimum distance
                const uint32_t bit = r.slot_bit_base[slot] + id;
                const uint32_t m = 1u << (bit & 31u);
                const uint32_t old = atomicOr(r.visited + (size_t)req * r.visited_words + (bit >> 5), m);
                if (old & m) active = false;
                else {
                    p = r.rprogs[slot];
                    first_visit = true;
                }
            } else {
                active = false;
            }
        }
        if (PHASE == REV_VISIT) {
            // seeds and first visits move on to the expand phase; states with parents elsewhere are also exported
            const uint4 o = make_uint4(id, req, first_visit ? (meta | kRevVisited) : meta, 0);
            const uint64_t b = __ballot(active);
            if (b) {
                const uint32_t base = reserve<false>(wo, (uint32_t)__popcll(b), lane);
                if (active && base != kNoSpace) out[base + lanes_below(b)] = o;
            }
            const bool xport = first_visit && (p.n & kRevRemoteBit) && dist < kMaxLevels;
            const uint4 xe = make_uint4(id, req, meta | kRevForeign, 0);
            if (!sh.by_dest) {
                export_entries(xport, xe, 0u, lane, sh);  // one block every shard receives; the importers keep what they hold parent rows for
            } else {
                // all-to-all form: a copy into the block of every shard that holds parent rows of this slot's states -- one round per distinct
                // destination among the lanes' lowest pending bits (a slot has one or two such shards; rounds = the wave's distinct destinations)
                uint64_t pend = xport ? r.rdest[slot] : 0ull;
                while (__ballot(pend != 0)) {
                    const uint32_t d = pend ? (uint32_t)(__ffsll((unsigned long long)pend) - 1) : 0u;
                    export_entries(pend != 0, xe, d, lane, sh);
                    pend &= pend - 1;
                }
            }
            continue;
        }
        const uint32_t nops = (active && dist < kMaxLevels) ? (p.n & ~kRevRemoteBit) : 0u;  // parents of a dist-50 state would need 51 levels
        uint32_t T = 0;
        const uint32_t maxops = uniform(wave_max(nops));
        for (uint32_t j = 0; j < maxops; j++) {
            bool want = false;
            uint32_t tstart = 0, tcount = 0, tmeta = 0;
            if (j < nops) {
                const RevOp op = r.rops[p.first + j];
                const bool dead = op.target < 256u && !((r.useful[op.target >> 5] >> (op.target & 31u)) & 1u);  // (cannot lead to the result slot: DevReverse::useful)
                if (dead) {
                } else if (op.flags & OP_PUSH_SAME) {
                    want = true;
                    tstart = id;
                    tcount = 1u | kSelfBit;
                } else if ((op.flags & OP_WILD) || id < op.nrows) {  // (OP_WILD, seeds only: the wildcard subject's row, whatever the seed's id)
                    const uint2 rd = reinterpret_cast<const uint2 *>(r.rmeta)[op.roff_base + ((op.flags & OP_WILD) ? 0u : id)];
                    const uint32_t s0 = rd.x, s1 = rd.y;
                    if (s1 - s0 > kMaxRow) *f.overflow = 2u;
                    else if (s1 > s0) {
                        want = true;
                        tstart = s0;
                        tcount = s1 - s0;
                    }
                }
                tmeta = op.target | ((dist + 1) << 13);
            }
            const uint64_t b = __ballot(want);
            if (b) {
                if (want) {
                    const uint32_t q = T + lanes_below(b);
                    t.a[q] = make_uint4(tstart, 0u, 1u, req);
                    t.count[q] = tcount;
                    t.meta[q] = tmeta;
                    t.sid[q] = 0;
                }
                T += (uint32_t)__popcll(b);
                if (T > kTaskCap - 64) {
                    flush_tasks<false, false, false>(t, T, wo, lane, nog, nullptr, nullptr, r.redges, nullptr, nullptr, sh);
                    T = 0;
                }
            }
        }
        if (T) flush_tasks<false, false, false>(t, T, wo, lane, nog, nullptr, nullptr, r.redges, nullptr, nullptr, sh);
        }
    }
    if (lane == 0) {
        if (wo.cur != kNoSpace) wo.cold->counts[wo.cur] = wo.fill;
        if (wo.produced) f.any[iter] = 1u;
    }
}

// ------------------------------------------------------- reverse walk, single launch
// LookupResources as the proxy issues it: one request per list call (reference pkg/authz/lookups.go:49-83), a few dozen in flight
// (one goroutine per request: responsefilterer.go:165-204).  The level loop above costs one k_rev_expand launch per reverse level plus a
// status round trip, whatever the batch size, and writes every result id through the frontier once more before the next launch sets
// its bit (the level that produces the ~10 k result pods of a C3 lookup is the children of ~100 namespace entries).  Here ONE launch
// answers the batch: block b walks lookup b through every reverse level by itself.
//   * entries of a level = true states (object id, slot); their distance is the level: uniform, not stored;
//   * phase A: every thread takes an entry and turns its program's ops into tasks (row start, degree, target) in an LDS list
//     (computed-userset parents -- PUSH_SAME -- are visited right there);
//   * phase B: a block-wide prefix sum over the degrees, then the 1024 lanes walk the OUTPUT index space: lane -> task by a binary
//     search over the LDS prefix array, consecutive lanes read consecutive resource ids of a reverse row (coalesced), four children
//     per lane in flight;
//   * a child is VISITED WHERE IT IS PRODUCED: test-and-set of its bit in the request's visited bitmap (first visit wins; everything
//     produced in one level has the same distance, so "first" is still "nearest" -- the depth-50 cut depends on it).  Children whose
//     slot has no parents (the lookup's own result slot, typically) need no answer from the atomic and are never written anywhere
//     else: the bit IS the result.  Only first visits of states that do have parents enter the block's private frontier region.
//   * the block zeroes its own bitmap at the start and, at the end, copies the result slot's words to the caller's rows (device,
//     or pinned host memory: no separate D2H copy) and counts the ids.
// A block that outgrows its region or meets a level of more than kRevLocalBudget children raises `overflow`: the host redoes the
// batch on the level loop, which spreads one huge lookup over the whole chip.
constexpr int kRevLocalThreads = 1024;
constexpr uint32_t kRevTaskCap = kRevLocalThreads;  // one (state, op) pair per thread and round => at most one task per thread
constexpr uint32_t kRevLocalBudget = 1u << 22;      // children of one round a single block may enumerate
constexpr uint32_t kRevTerminal = 0x80000000u;      // task target flag: children are only marked, never expanded
constexpr uint32_t kRevNoMark = 0x40000000u;        // task target flag (OP_NOMARK seed ops): every child is a first visit by construction -- no visited bit
struct RevTaskLds {
    uint32_t start[kRevTaskCap];       // first resource id of the row in `redges`
    uint32_t prefix[kRevTaskCap + 1];  // exclusive prefix of the degrees (a thread without a task: degree 0)
    uint32_t target[kRevTaskCap];      // child slot | kRevTerminal
};
struct RevProgLds {  // the reverse programs, staged once per block (the host only takes this path when they fit)
    RevOp ops[kRevLdsOps];
    RevProg progs[kRevLdsSlots];
    uint2 slot[kRevLdsSlots];  // {first bit of the slot's visited rows, id space they cover}
};

// A small batch's caller spins on `done_flag` (pinned host memory) instead of synchronising the stream (see k_check_local): every block releases
// what it stored into host memory at system scope, arrives at the device counter, and the last arrival re-arms the counter and raises the flag.
__device__ __forceinline__ void signal_done(uint32_t *done_ctr, uint32_t *done_flag, uint32_t done_val) {
    if (!done_flag) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        if (__hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done_flag, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Round 6 -- BIG result types (rows beyond the block's LDS: the 8.45 M pods of BASELINE configs[4]'s graph, a 1 MB row).  One block enumerated every child of
// such a lookup -- 100-200 k pods under a few hundred namespaces and groups -- at ~5.5 ns each: 0.65-1.2 ms per lookup (profiles/r06_lookup_big.txt), and the
// level loop, which spreads a lookup over the chip, pays ~80 us per level for it.  What makes these lookups heavy is almost entirely TERMINAL: children in the
// lookup's own result slot, which are marked and never expanded -- no first-visit answer is needed, no order, no frontier.  So with `dfr.tasks` given the block
// does not enumerate such rows when a round holds `min_children` children or more: it appends {first edge, degree} to the lookup's DEFERRED list and walks on;
//   k_rev_terminal  (the next launch on the stream, blocks all over the chip) marks the deferred rows' ids in the lookup's result row -- a kernel boundary
//                   later, so the block's workgroup-scope marks are in memory and agent-scope atomics from every XCD meet them there;
//   k_rev_rows      (the launch after) copies the rows out slice by slice from all CUs, counts the ids, clears the slices it copied -- the three things that
//                   cost one block 80 us per lookup -- and the last block to finish hands the counts over and raises the caller's completion word.
struct RevDefer {
    uint2 *tasks = nullptr;      // [lookups][cap] {first edge, degree}; nullptr: nothing is deferred and the block copies its row itself (rows in LDS)
    uint32_t *count = nullptr;   // [lookups] tasks appended (zeroed by the block)
    uint32_t *levels = nullptr;  // [lookups] reverse levels walked (statistics; k_rev_rows packs them into the counts)
    uint32_t cap = 0;
    uint32_t min_children = 4096;  // children of one round from which its terminal rows go to the chip-wide launch (below: the block is done with them sooner than a launch starts)
    // The marks themselves are BYTES, not bits: one byte per object of the result type, [lookups][bm_stride], zero between launches.  Marking 100-200 k ids with
    // atomic ORs -- workgroup scope from one CU (round 5) or agent scope from all of them (this round's first version) -- ran at ~5 ns per id either way
    // (profiles/r06_lookup_big.txt): the read-modify-write is the cost.  A result slot nobody expands needs no answer from the mark, so a plain byte store does
    // (idempotent, no neighbour to lose); k_rev_rows folds 32 bytes into a word of the caller's row and zeroes the bytes it found set.
    uint8_t *bytemap = nullptr;
    uint32_t bm_stride = 0;
};

__global__ __launch_bounds__(kRevLocalThreads) void k_rev_local(DevReverse r, const uint32_t *__restrict__ sids, uint32_t key, uint32_t target_in,
                                                                 uint2 *buf0, uint2 *buf1, uint32_t cap, uint32_t *out_bitmaps, uint32_t out_stride,
                                                                 uint32_t copy_words, unsigned long long *out_counts, uint32_t *status, uint32_t lds_words, uint32_t *done_ctr,
                                                                 uint32_t *done_flag, uint32_t done_val, RevDefer dfr, RevUseful useful) {
    // (the result slot, and whether it is a SINK of the reverse graph -- Snapshot::rev_sink: no parent of its states can make another of its states true, so they are
    //  marked and not expanded: a lookup of `edit` does not walk on into `view = viewer + edit`, a lookup of namespace#view not into the namespaces' pods)
    const uint32_t target_slot = target_in & ~kRevTargetSink;
    const bool target_sink = (target_in & kRevTargetSink) != 0u;
    __shared__ RevTaskLds t;
    __shared__ RevProgLds pl;
    // lds_words != 0: the RESULT slot's rows live here, not in `visited` -- the level that produces a lookup's ids (thousands of pods under
    // a few hundred namespaces) then marks them with LDS atomics instead of L2 atomics, and the result row is copied out of LDS.  The
    // launcher sizes it to the slot's id space (up to 128 KiB = 1 M objects; beyond that, lds_words == 0 and the rows stay in HBM).
    extern __shared__ uint32_t s_row[];
    // The block's frontier is ONE append-only log (buf0's region; buf1 is unused): level L reads [lvl_lo, lvl_hi) and appends behind the end.
    // Every entry is a first visit whose bit this block set in `visited` -- so the log is also the list of what to clear afterwards (below).
    __shared__ uint32_t s_end, s_wave_tot[kRevLocalThreads / 64], s_stop, s_maxops, s_count[kRevLocalThreads / 64];
    __shared__ uint32_t s_nomark[kRevLdsSlots / 32];  // slots whose states carry no visited bit in this launch (nothing to clear afterwards)
    const uint32_t tid = threadIdx.x, lane = lane_id(), wib = tid >> 6;
    const uint32_t req = blockIdx.x;
    uint32_t *__restrict__ visited = r.visited + (size_t)req * r.visited_words;
    uint2 *const log = buf0 + (size_t)req * cap;
    (void)buf1;
    const uint2 *__restrict__ rmeta2 = reinterpret_cast<const uint2 *>(r.rmeta);
    const uint32_t *__restrict__ redges = r.redges;
    __shared__ uint32_t s_dfr;  // tasks this block has deferred
    if (tid == 0) {
        s_end = 0;
        s_stop = 0;
        s_maxops = 0;
        s_dfr = 0;
    }
    if (tid < kRevLdsSlots / 32) s_nomark[tid] = 0;
    __syncthreads();
    // (ops whose target slot cannot lead to the result slot are dead for this lookup: Snapshot::rev_useful -- a lookup of pod#creator does not walk the user's groups)
    for (uint32_t i = tid; i < r.nrops; i += kRevLocalThreads) {
        RevOp o = r.rops[i];
        if (o.target < kRevLdsSlots && !((useful.w[o.target >> 5] >> (o.target & 31u)) & 1u)) o.flags |= OP_DEAD;
        pl.ops[i] = o;
    }
    for (uint32_t i = tid; i < r.nslots; i += kRevLocalThreads) {
        const RevProg p = r.rprogs[i];
        pl.progs[i] = p;
        pl.slot[i] = make_uint2(r.slot_bit_base[i], r.slot_nobjects[i]);
        atomicMax(&s_maxops, p.n & ~kRevRemoteBit);
    }
    const RevProg seed = r.rseeds[key];
    const uint32_t sid = sids[req];
    const uint32_t row_w0 = r.slot_bit_base[target_slot] >> 5;  // first word of the result slot's rows in `visited`
    for (uint32_t i = tid; i < (seed.n & ~kRevRemoteBit); i += kRevLocalThreads) {
        const RevOp so = r.rops[seed.first + i];
        if ((so.flags & OP_NOMARK) && so.target != target_slot) atomicOr(&s_nomark[so.target >> 5], 1u << (so.target & 31u));
    }
    for (uint32_t i = tid; i < lds_words; i += kRevLocalThreads) s_row[i] = 0u;
    // `visited` is NOT zeroed here (round 4; VERDICT r3 weak #3: 64 blocks zeroing ~50 KB each were 7x the result rows in write traffic):
    // the host hands it over all-zero once, and every block clears exactly the bits it set before it ends -- the non-terminal first visits
    // are all in its log, marks of the result slot are cleared with the row, and terminal states of OTHER slots are not marked at all
    // (nobody reads those bits: they are not expanded and not part of the answer).
    // The bitmap is private to this block: WORKGROUP scope everywhere.  (Agent scope here was the kernel's whole cost beyond 16 lookups: an
    // agent-scope atomic on gfx950 is a fabric transaction that drops the line from the L2, and __threadfence() walks the L2's dirty lines.)
    __syncthreads();

    // marks child (slot, id); returns true when it was a first visit of a state that has parents of its own
    auto visit = [&](uint32_t id, uint32_t tgt, bool valid) -> bool {
        const uint32_t slot = tgt & ~(kRevTerminal | kRevNoMark);
        const uint2 si = pl.slot[slot];
        const bool ok = valid && id < si.y;
        bool push = false;
        if (slot == target_slot) {
            if (ok && lds_words) {
                const uint32_t m = 1u << (id & 31u);
                if (tgt & kRevTerminal) (void)atomicOr(&s_row[id >> 5], m);
                else push = !(atomicOr(&s_row[id >> 5], m) & m);
                return push;
            }
            if (dfr.bytemap) {  // (rows in HBM, result slot terminal: the mark is the object's byte)
                if (ok) dfr.bytemap[(size_t)req * dfr.bm_stride + id] = 1;
                return false;
            }
        } else if (tgt & kRevTerminal) {
            return false;  // neither expanded nor part of the answer: no bit
        } else if (tgt & kRevNoMark) {
            return ok;     // the only producer of this slot's states, each id once: a first visit without asking
        }
        const uint32_t bit = si.x + (ok ? id : 0u);
        uint32_t *w = visited + (bit >> 5);
        const uint32_t m = 1u << (bit & 31u);
        if (ok) {
            if (tgt & kRevTerminal) (void)__hip_atomic_fetch_or(w, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (result unused: no return trip)
            else push = !(__hip_atomic_fetch_or(w, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & m);
        }
        return push;
    };
    // wave-cooperative append to the block's log (call in wave-uniform control flow)
    auto append = [&](bool push, uint32_t id, uint32_t slot) {
        const uint64_t b = __ballot(push);
        if (!b) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_end, (uint32_t)__popcll(b));
        base = uniform(base);
        if (base + (uint32_t)__popcll(b) > cap) {
            if (lane == 0) s_stop = 1u;
            return;
        }
        if (push) log[base + lanes_below(b)] = make_uint2(id, slot);
    };

    uint32_t lvl_lo = 0, cnt = 1, level = 1;
    int stop = 0;  // block-uniform copy of s_stop (taken through a barrier: s_stop itself may be raised by a faster wave at any time)
    for (; level <= kMaxLevels; level++) {  // level L: states at distance L - 1 (all of them: uniform) produce children at distance L
        const uint2 *__restrict__ fin = log + lvl_lo;
        const uint32_t lvl_hi = lvl_lo + (level > 1 ? cnt : 0u);  // (the seed is not in the log)
        // a child at distance 50 is marked but never expanded (its parents would sit at 51): terminal whatever its slot
        const uint32_t term_all = level >= kMaxLevels ? kRevTerminal : 0u;
        // work items of a level = (state, op) pairs, one per thread and round: a seed's six rows, or the one or two parent ops of a
        // few hundred states, are fetched side by side instead of one after the other by the state's thread
        const uint32_t W = level == 1 ? (seed.n & ~kRevRemoteBit) : s_maxops;
        const uint32_t npairs = cnt * W;
        for (uint32_t pb = 0; pb < npairs; pb += kRevLocalThreads) {
            // ---- phase A: this thread's op -> at most one task (start, degree, target), kept in registers
            const uint32_t q = pb + tid;
            uint32_t deg = 0, start = 0, tgt = 0;
            bool same = false;  // computed-userset parent: the child is the same object
            uint32_t id = 0;
            if (q < npairs) {
                const uint32_t e = q / W, j = q - e * W;
                RevProg p = seed;
                id = sid;
                if (level > 1) {
                    const uint2 en = fin[e];
                    id = en.x;
                    p = pl.progs[en.y];
                }
                if (j < (p.n & ~kRevRemoteBit) && !(pl.ops[p.first + j].flags & OP_DEAD)) {
                    const RevOp op = pl.ops[p.first + j];
                    uint32_t np = pl.progs[op.target].n & ~kRevRemoteBit;
                    if (target_sink && op.target == target_slot) np = 0u;
                    tgt = op.target | (np == 0u ? kRevTerminal : 0u) | term_all | ((op.flags & OP_NOMARK) ? kRevNoMark : 0u);
                    // Round 6: a row whose children land in a relation that does nothing but feed ONE permission nobody expands -- `pod#viewer`, whose only
                    // parent is the computed userset in `pod#view = viewer + ...` -- marks that permission's objects directly, one dispatch level further
                    // on (so only while that level is still inside the depth limit).  The intermediate states were first-visited, logged and pushed
                    // through a level of their own, 1 024 per round: half of a big lookup's ids took that detour at ~5 ns each (profiles/r06_lookup_big.txt),
                    // and it is what kept their rows out of the deferred list.  Nobody reads the intermediate slot's bits unless it is the result slot.
                    if (!(op.flags & OP_PUSH_SAME) && np == 1u && op.target != target_slot) {
                        const RevOp via = pl.ops[pl.progs[op.target].first];
                        if ((via.flags & OP_PUSH_SAME) && ((pl.progs[via.target].n & ~kRevRemoteBit) == 0u || (target_sink && via.target == target_slot))) {
                            if (level + 1u <= kMaxLevels) tgt = via.target | kRevTerminal;
                            else np = 0xFFFFFFFFu;  // (the permission's level lies beyond the limit: the relation's states are marked -- nobody asks -- and end there)
                        }
                    }
                    if (np == 0xFFFFFFFFu) {
                        // nothing to enumerate for this op
                    } else if (op.flags & OP_PUSH_SAME) {
                        same = true;
                    } else if ((op.flags & OP_WILD) || id < op.nrows) {  // (OP_WILD, seeds only: the wildcard subject's row, whatever the seed's id)
                        const uint2 rd = rmeta2[op.roff_base + ((op.flags & OP_WILD) ? 0u : id)];
                        if (rd.y - rd.x > kMaxRow) s_stop = 2u;
                        else if (rd.y > rd.x) {
                            start = rd.x;
                            deg = rd.y - rd.x;
                        }
                    }
                }
            }
            {  // same-object parents are visited right here (wave-uniform control flow: the append ballots)
                const bool push = visit(id, tgt, same);
                append(push, id, tgt & ~(kRevTerminal | kRevNoMark));
            }
            // ---- block-wide exclusive prefix of the degrees
            const uint32_t incl = wave_incl_scan(deg, lane);
            if (lane == 63) s_wave_tot[wib] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < kRevLocalThreads / 64; w++) {
                const uint32_t wt = s_wave_tot[w];
                before += w < wib ? wt : 0u;
                total += wt;
            }
            if (dfr.tasks && total >= dfr.min_children) {  // (block-uniform) a heavy round: its terminal rows of the result slot go to the chip-wide launch
                const bool mine = deg != 0u && (tgt & kRevTerminal) && (tgt & ~(kRevTerminal | kRevNoMark)) == target_slot;
                const uint64_t b = __ballot(mine);
                if (b) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_dfr, (uint32_t)__popcll(b));
                    base = uniform(base);
                    if (base + (uint32_t)__popcll(b) > dfr.cap) {
                        if (lane == 0) s_stop = 1u;  // (more rows than the list holds: the level loop takes the batch)
                    } else if (mine) {
                        dfr.tasks[(size_t)req * dfr.cap + base + lanes_below(b)] = make_uint2(start, deg);
                        deg = 0;
                    }
                }
                __syncthreads();  // (every wave has read the totals above)
                const uint32_t incl2 = wave_incl_scan(deg, lane);
                if (lane == 63) s_wave_tot[wib] = incl2;
                __syncthreads();
                before = 0;
                total = 0;
#pragma unroll
                for (uint32_t w = 0; w < kRevLocalThreads / 64; w++) {
                    const uint32_t wt = s_wave_tot[w];
                    before += w < wib ? wt : 0u;
                    total += wt;
                }
                before += incl2 - incl;  // (the code below takes this thread's exclusive prefix as before + incl - deg)
            }
            if (total) {  // (block-uniform)
                t.prefix[tid] = before + incl - deg;
                t.start[tid] = start;
                t.target[tid] = tgt;
                if (tid == 0 && total > kRevLocalBudget) s_stop = 1u;  // one block should not enumerate this alone: the level loop takes the batch
                __syncthreads();
                // ---- phase B: the lanes walk the output index space; lane -> task by binary search (threads without a task have degree 0:
                // "the last task whose prefix is <= w" is always the one that owns w)
                if (total <= kRevLocalBudget) {
                    constexpr int U = 2;  // children per lane in flight (round 5, same-box A/B: a power-user lookup 20.3 us with 4, 19.0 with 2, 22.7 with 8)
                    for (uint32_t wb = 0; wb < total; wb += U * kRevLocalThreads) {  // (block-uniform trip count)
                        uint32_t edge[U], tj[U];
                        bool valid[U];
#pragma unroll
                        for (int k = 0; k < U; k++) {
                            const uint32_t w = wb + (uint32_t)k * kRevLocalThreads + tid;
                            valid[k] = w < total;
                            const uint32_t wv = valid[k] ? w : total - 1;  // inactive lanes shadow the last child: every load stays in range
                            uint32_t jt = 0;
#pragma unroll
                            for (uint32_t step = kRevTaskCap / 2; step >= 1; step >>= 1)
                                if (t.prefix[jt + step] <= wv) jt += step;
                            tj[k] = jt;
                            edge[k] = gld(redges, t.start[jt] + (wv - t.prefix[jt]));
                        }
                        issue_fence();  // the U gathers travel together
#pragma unroll
                        for (int k = 0; k < U; k++) {
                            if (!__ballot(valid[k])) break;  // (wave-uniform)
                            const uint32_t tg = t.target[tj[k]];
                            const bool push = visit(edge[k], tg, valid[k]);
                            append(push, edge[k], tg & ~(kRevTerminal | kRevNoMark));
                        }
                    }
                }
            }
            // (also the barrier behind phase B: every lane is done with the task list before the next round overwrites it)
            stop = __syncthreads_or(s_stop != 0u);
            if (stop) break;
        }
        if (stop) break;
        const uint32_t end = s_end;  // (every append of this level is behind the round's closing barrier; <= cap, or `stop` were set)
        if (end == lvl_hi) break;   // nothing produced
        lvl_lo = lvl_hi;
        cnt = end - lvl_hi;
    }
    if (stop) {
        // 1: redo on the level loop, 2: a row beyond the per-task enumeration limit.  A plain store (the flag may live in pinned host
        // memory): blocks that race write non-zero either way, and a 2 lost to a 1 is found again by the level loop.
        // (`visited` is left dirty: the host zeroes it before the next single-launch lookup on this context)
        if (tid == 0) {
            *status = s_stop;
            if (dfr.tasks) {  // (the launches behind this one find nothing to do for this lookup; the host redoes the batch)
                dfr.count[req] = 0u;
                dfr.levels[req] = 0u;
            }
        }
        if (!dfr.tasks) signal_done(done_ctr, done_flag, done_val);  // (deferral: k_rev_rows raises the completion word, behind everything)
        return;
    }
    if (dfr.tasks) {
        // ---- rows in HBM, handled by the two launches behind this one (k_rev_terminal, k_rev_rows): what is left here is to leave `visited` as it was
        // handed over OUTSIDE the result row -- the logged first visits -- and to publish how much was deferred
        __syncthreads();
        const uint32_t nlog = min(s_end, cap);
        for (uint32_t i = tid; i < nlog; i += kRevLocalThreads) {
            const uint2 en = log[i];
            if (en.y == target_slot) continue;  // (the row is cleared by k_rev_rows, behind the copy)
            if (s_nomark[en.y >> 5] >> (en.y & 31u) & 1u) continue;
            visited[(pl.slot[en.y].x + en.x) >> 5] = 0u;
        }
        if (tid == 0) {
            dfr.count[req] = min(s_dfr, dfr.cap);
            dfr.levels[req] = min(level, kMaxLevels);
        }
        return;
    }
    // ---- result rows: the target slot's words (every one of them was last written by an L2 atomic or is still the zero it was handed over
    // with; the loads below are served by that L2 -- sc1 loads bypass the vector L1, which atomics never update)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    uint32_t *orow = out_bitmaps + (size_t)req * out_stride;
    uint32_t c32 = 0;  // (a row holds < 2^31 ids)
    for (uint32_t i = tid; i < out_stride; i += kRevLocalThreads) {
        uint32_t v = 0;
        if (i < copy_words) v = lds_words ? s_row[i] : __hip_atomic_load(visited + row_w0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        orow[i] = v;
        c32 += (uint32_t)__popc(v);
    }
    if (out_counts) {
        c32 = wave_last(wave_incl_scan(c32, lane));
        if (lane == 0) s_count[wib] = c32;
        __syncthreads();
        if (tid == 0) {
            unsigned long long tot = 0;
            for (uint32_t w = 0; w < kRevLocalThreads / 64; w++) tot += s_count[w];
            out_counts[req] = tot | ((unsigned long long)min(level, kMaxLevels) << 56);  // + the reverse levels this lookup walked (statistics)
        }
    }
    // ---- leave `visited` as it was handed over: all zero.  Every bit this block set outside the result row belongs to a logged first visit
    // (whole words: they hold this lookup's bits only); the result row, where it lives in HBM, goes as a range.
    __syncthreads();  // (the row above is copied out before anything in it is cleared)
    const uint32_t nlog = min(s_end, cap);
    for (uint32_t i = tid; i < nlog; i += kRevLocalThreads) {
        const uint2 en = log[i];
        if (lds_words && en.y == target_slot) continue;  // (marked in LDS)
        if (s_nomark[en.y >> 5] >> (en.y & 31u) & 1u) continue;  // (never marked)
        visited[(pl.slot[en.y].x + en.x) >> 5] = 0u;
    }
    if (!lds_words) {
        const uint32_t rw = (pl.slot[target_slot].y + 31u) >> 5;
        for (uint32_t i = tid; i < rw; i += kRevLocalThreads) visited[row_w0 + i] = 0u;
    }
    signal_done(done_ctr, done_flag, done_val);
}

// The deferred rows of k_rev_local (see RevDefer): blockIdx.y = lookup, the launch's waves take its tasks round-robin and a wave's lanes read consecutive ids
// of a reverse row.  An id's mark is its byte of the lookup's byte map (RevDefer::bytemap): a plain store, whichever block and XCD makes it.
__global__ __launch_bounds__(256) void k_rev_terminal(DevReverse r, uint32_t target_slot, RevDefer dfr) {
    // (a lookup whose block gave up has published zero tasks; the status word itself lives in the caller's pinned memory and is not worth a trip across PCIe
    //  from every thread of a chip-wide launch -- a first version read it and spent 100 us of a 220 us empty lookup doing so)
    const uint32_t req = blockIdx.y, lane = lane_id();
    const uint32_t nt = min(dfr.count[req], dfr.cap), nobj = r.slot_nobjects[target_slot];
    uint8_t *__restrict__ bm = dfr.bytemap + (size_t)req * dfr.bm_stride;
    const uint2 *__restrict__ tasks = dfr.tasks + (size_t)req * dfr.cap;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    for (uint32_t t = wave; t < nt; t += nwaves) {
        const uint2 tk = tasks[t];
        for (uint32_t i = lane; i < tk.y; i += 64u) {
            const uint32_t id = gld(r.redges, tk.x + i);
            if (id < nobj) bm[id] = 1;
        }
    }
}

// The result rows of a batch whose rows live in HBM (see RevDefer): blockIdx.y = lookup, the x blocks take slices of its row -- copy out (device or pinned
// host memory), count, clear -- so that a 1 MB row crosses PCIe from all CUs at once instead of from one.  d_count: [lookups] device accumulators (zero
// between launches: re-armed by the last block); the last block of the LAUNCH writes every lookup's count | levels << 56 to out_counts and raises done_flag.
__global__ __launch_bounds__(256) void k_rev_rows(DevReverse r, uint32_t target_slot, uint32_t n, uint32_t *out_bitmaps, uint32_t out_stride, uint32_t copy_words,
                                                  unsigned long long *out_counts, unsigned long long *d_count, RevDefer dfr, uint32_t *done_ctr,
                                                  uint32_t *done_flag, uint32_t done_val) {
    __shared__ uint32_t s_cnt[4];
    __shared__ bool s_last;
    const uint32_t req = blockIdx.y, lane = lane_id(), wib = threadIdx.x >> 6;
    // (a batch in which some block gave up is redone by the host on the level loop: copying its partial rows is harmless, and the host zeroes the byte map again)
    uint4 *__restrict__ bm = reinterpret_cast<uint4 *>(dfr.bytemap + (size_t)req * dfr.bm_stride);  // (bm_stride is a multiple of 128 bytes: 32 ids = two aligned uint4)
    uint32_t *__restrict__ orow = out_bitmaps + (size_t)req * out_stride;
    uint32_t c32 = 0;
    auto nibble = [](uint32_t x) -> uint32_t { return (x * 0x01020408u) >> 24 & 0xFu; };  // four 0 / 1 bytes -> four bits, byte 0 lowest (no two partial products meet)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < out_stride; i += gridDim.x * 256u) {  // one word of the caller's row = 32 bytes of the map
        uint32_t v = 0;
        if (i < copy_words) {
            const uint4 a = bm[2u * i], b = bm[2u * i + 1u];
            v = nibble(a.x) | nibble(a.y) << 4 | nibble(a.z) << 8 | nibble(a.w) << 12 | nibble(b.x) << 16 | nibble(b.y) << 20 | nibble(b.z) << 24 | nibble(b.w) << 28;
            if (v) {  // (zero again for the next lookup: only where something was set)
                bm[2u * i] = make_uint4(0u, 0u, 0u, 0u);
                bm[2u * i + 1u] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        orow[i] = v;
        c32 += (uint32_t)__popc(v);
    }
    c32 = wave_last(wave_incl_scan(c32, lane));
    if (lane == 0) s_cnt[wib] = c32;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (tot) (void)__hip_atomic_fetch_add(d_count + req, (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();  // this block's slices are in the caller's memory before it arrives
        s_last = __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) {
        const unsigned long long c = __hip_atomic_exchange(d_count + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (out_counts) out_counts[i] = c | ((unsigned long long)dfr.levels[i] << 56);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done_flag) __hip_atomic_store(done_flag, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------ import
// Appends entries of an exchanged export buffer to the frontier iteration `iter` produced (dynamic chunks only:
// the static chunks belong to that iteration's expand waves).  FWD: keep the entries whose slot this shard owns.
// Reverse: keep foreign states for which this shard holds parent rows.
template <bool FWD>
__global__ __launch_bounds__(256) void k_import(DevFrontier f, uint32_t iter, const uint4 *__restrict__ in, uint32_t n, const SlotProg *progs,
                                                const RevProg *rprogs, uint32_t rank) {
    const uint32_t lane = lane_id();
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    uint4 *__restrict__ out = f.buf[iter & 1u];
    uint32_t *out_counts = f.counts[iter & 1u];
    uint32_t *out_nchunks = f.nchunks + iter;
    uint32_t cur = kNoSpace, fill = kChunk, produced = 0;
    for (uint32_t x = wave; (uint64_t)x * 64 < n; x += nw) {
        const uint32_t i = x * 64 + lane;
        const uint4 e = i < n ? in[i] : make_uint4(0, 0, kDeadMeta, 0);
        bool mine = i < n && e.z != kDeadMeta;
        if (mine) mine = FWD ? progs[meta_slot(e.z)].owner == rank : (rprogs[e.z & 0x1FFFu].n & ~kRevRemoteBit) != 0;
        const uint64_t b = __ballot(mine);
        if (!b) continue;
        const uint32_t need = (uint32_t)__popcll(b);
        if (fill + need > kChunk) {
            if (lane == 0 && cur != kNoSpace) out_counts[cur] = fill;
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(out_nchunks, 1u);
            c = uniform(c) + f.nwaves;
            if (c >= f.max_chunks) {
                if (lane == 0) *f.overflow = 1u;
                cur = kNoSpace;
                break;
            }
            cur = c;
            fill = 0;
        }
        if (mine) out[(size_t)cur * kChunk + fill + lanes_below(b)] = e;
        fill += need;
        produced += need;
    }
    if (lane == 0) {
        if (cur != kNoSpace) out_counts[cur] = fill;
        if (produced) f.any[iter] = 1u;
    }
}

// ---------------------------------------------------------------- native sharded loop (engine_shard_native.cpp)
// Exchange blocks, one per (shard, level) in the all-gather form and one per (shard, destination, level) in the all-to-all form: a 16-byte
// HEADER and up to `cap` entries.  header = {x: entries in this block, y: this shard produced local work | overflow code << 1,
// z: the largest block this shard filled this level, w: entries this shard exported this level in all}.  Written on the device, so the host
// never waits between levels; headers and entries travel in separate collectives (the entries' one is skipped on levels that are known
// to export nothing).  exp_count = [0] all-gather form; [1 + d] per destination.
__global__ __launch_bounds__(64) void k_xhdr(uint4 *hdr, uint32_t nblocks, const uint32_t *exp_count, const uint32_t *any_iter, const uint32_t *overflow) {
    const uint32_t d = threadIdx.x;
    if (d >= nblocks) return;
    const uint32_t flags = (*any_iter ? 1u : 0u) | (*overflow << 1);
    if (nblocks == 1) {
        const uint32_t c = exp_count[0];
        hdr[0] = make_uint4(c, flags, c, c);
        return;
    }
    uint32_t mx = 0, tot = 0;
    for (uint32_t k = 0; k < nblocks; k++) {
        const uint32_t c = exp_count[1 + k];
        mx = max(mx, c);
        tot += c;
    }
    hdr[d] = make_uint4(exp_count[1 + d], flags, mx, tot);
}

// After the exchange: blockIdx.y = source shard.  Rows of other shards import what they sent (FWD: the entries whose slot this shard owns;
// reverse: foreign states for which this shard holds parent rows), the count read from the source's header; the own row's first wave folds all
// headers into the level's control record {total exported, any shard produced, overflow code, largest block} -- identical on every shard, so
// all of them take the same decisions (done / redo) without talking to each other or to the host.  have_data == 0: the entries were not
// exchanged this level (planned: nothing was expected); headers that announce entries then raise overflow code 4 = "redo with the data".
template <bool FWD>
__global__ __launch_bounds__(256) void k_import_gathered(DevFrontier f, uint32_t iter, const uint4 *__restrict__ hdrs, const uint4 *__restrict__ data, uint32_t world,
                                                         uint32_t rank, uint32_t cap, uint32_t have_data, const SlotProg *progs, const RevProg *rprogs,
                                                         uint32_t *ctrl) {
    const uint32_t lane = lane_id();
    const uint32_t src = blockIdx.y;
    if (src == rank) {
        if (blockIdx.x == 0 && threadIdx.x < 64) {
            uint32_t total = 0, anyp = 0, over = 0, mx = 0;
            for (uint32_t r = lane; r < world; r += 64) {
                const uint4 h = hdrs[r];
                total += h.w;
                anyp |= h.y & 1u;
                over |= (h.y >> 1) | (h.z > cap ? 1u : 0u) | ((!have_data && h.w) ? 4u : 0u);  // (h.w, every shard's own included: the same verdict on every shard)
                mx = max(mx, h.z);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                total += (uint32_t)__shfl_xor((int)total, d, 64);
                anyp |= (uint32_t)__shfl_xor((int)anyp, d, 64);
                over |= (uint32_t)__shfl_xor((int)over, d, 64);
                mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
            }
            if (lane == 0) {
                ctrl[0] = total;
                ctrl[1] = anyp;
                ctrl[2] = over;
                ctrl[3] = mx;
            }
        }
        return;
    }
    if (!have_data) return;
    const uint4 *__restrict__ in = data + (size_t)src * cap;
    const uint32_t n = min(hdrs[src].x, cap);
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    uint4 *__restrict__ out = f.buf[iter & 1u];
    uint32_t *out_counts = f.counts[iter & 1u];
    uint32_t *out_nchunks = f.nchunks + iter;
    uint32_t cur = kNoSpace, fill = kChunk, produced = 0;
    for (uint32_t x = wave; (uint64_t)x * 64 < n; x += nw) {
        const uint32_t i = x * 64 + lane;
        const uint4 e = i < n ? in[i] : make_uint4(0, 0, kDeadMeta, 0);
        bool mine = i < n && e.z != kDeadMeta;
        if (mine) mine = FWD ? progs[meta_slot(e.z)].owner == rank : (rprogs[e.z & 0x1FFFu].n & ~kRevRemoteBit) != 0;
        const uint64_t b = __ballot(mine);
        if (!b) continue;
        const uint32_t need = (uint32_t)__popcll(b);
        if (fill + need > kChunk) {
            if (lane == 0 && cur != kNoSpace) out_counts[cur] = fill;
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(out_nchunks, 1u);
            c = uniform(c) + f.nwaves;
            if (c >= f.max_chunks) {
                if (lane == 0) *f.overflow = 1u;
                cur = kNoSpace;
                break;
            }
            cur = c;
            fill = 0;
        }
        if (mine) out[(size_t)cur * kChunk + fill + lanes_below(b)] = e;
        fill += need;
        produced += need;
    }
    if (lane == 0) {
        if (cur != kNoSpace) out_counts[cur] = fill;
        if (produced) f.any[iter] = 1u;
    }
}

}  // namespace

#if ACL_PROFILE_PHASES
}  // namespace acl
extern "C" int acl_debug_phase_cycles(unsigned long long *out16) {  // variant builds only (tools/phases.sh): reads and resets the counters
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(acl::acl_phase_cycles), sizeof(z)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(acl::acl_phase_cycles), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
namespace acl {
#endif
static size_t prog_lds_bytes(const DevGraph &g) { return ((size_t)g.nslots + g.nops) * 32; }
// A/B knob: ACL_PROG_LDS=0 runs the instantiations that read the program table from global memory (8 KiB less LDS per block)
static bool prog_in_lds() {
    static const bool on = [] {
        const char *e = getenv("ACL_PROG_LDS");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

int expand_grid_blocks(int device) {
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    int per_cu = 8;  // 256-thread blocks, <= 64 VGPRs, ~20 KiB LDS
    if (const char *e = getenv("ACL_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;  // A/B knob (tools/ab.sh)
    int occ = 0;
    // (the program table's LDS copy is dynamic; 2 KiB covers schemas of ~60 slots + ops -- a larger one only means some blocks of a launch queue)
    const hipError_t oe = prog_in_lds() ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_expand<true, false>, kBlock, 2048)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_expand<false, false>, kBlock, 0);
    if (oe == hipSuccess && occ > 0) per_cu = occ < per_cu ? occ : per_cu;
    return cus * per_cu;
}

void launch_seed(hipStream_t s, const DevGraph &g, const DevFrontier &f, const uint4 *items, uint32_t n, uint8_t *has, uint8_t *err, const DevShard &sh) {
    const uint32_t threads = std::max(std::max(n, f.nwaves), kStatusWords);
    hipLaunchKernelGGL(k_seed, dim3((threads + 255) / 256), dim3(256), 0, s, g, f, items, n, has, err, sh);
}
void launch_expand(hipStream_t s, const DevGraph &g, const DevFrontier &f, uint32_t iter, uint8_t *has, uint8_t *err, const DevShard &sh) {
    const dim3 grid(f.nwaves / kWavesPerBlock);
    const bool lds = g.nslots + g.nops <= kProgLdsEntries && prog_in_lds();
    if (g.bexpr) {  // schemas with `&` / `-`
        if (sh.world > 1) {  // (round 5: the sharded graph too -- cells in per-shard ranges of one global cell space, engine_shard_native.cpp)
            if (lds) hipLaunchKernelGGL((k_expand<true, true, true>), grid, dim3(kBlock), prog_lds_bytes(g), s, g, f, iter, has, err, sh);
            else hipLaunchKernelGGL((k_expand<false, true, true>), grid, dim3(kBlock), 0, s, g, f, iter, has, err, sh);
            return;
        }
        if (lds) hipLaunchKernelGGL((k_expand<true, false, true>), grid, dim3(kBlock), prog_lds_bytes(g), s, g, f, iter, has, err, sh);
        else hipLaunchKernelGGL((k_expand<false, false, true>), grid, dim3(kBlock), 0, s, g, f, iter, has, err, sh);
        return;
    }
    if (sh.world > 1) {
        if (lds) hipLaunchKernelGGL((k_expand<true, true>), grid, dim3(kBlock), prog_lds_bytes(g), s, g, f, iter, has, err, sh);
        else hipLaunchKernelGGL((k_expand<false, true>), grid, dim3(kBlock), 0, s, g, f, iter, has, err, sh);
    } else {
        if (lds) hipLaunchKernelGGL((k_expand<true, false>), grid, dim3(kBlock), prog_lds_bytes(g), s, g, f, iter, has, err, sh);
        else hipLaunchKernelGGL((k_expand<false, false>), grid, dim3(kBlock), 0, s, g, f, iter, has, err, sh);
    }
}
template <int WAVES>
static void launch_check_local_w(hipStream_t s, const DevGraph &g, const uint4 *items, uint32_t n, uint32_t rpw, uint32_t nblocks, uint32_t nunits, uint32_t nstatic, uint32_t rdyn,
                                 uint32_t *next_unit, uint4 *buf0, uint4 *buf1, uint32_t cap, uint32_t *overflow, uint8_t *has, uint8_t *err, uint8_t *perm_out, int32_t *err_out,
                                 uint32_t *max_level, uint32_t skew, uint32_t *done_ctr, uint32_t *done_flag, uint32_t done_val, const InlineItems &inl) {
    const dim3 grid(nblocks);
    const bool lds = g.nslots + g.nops <= kProgLdsEntries && prog_in_lds();
    if (g.bexpr) {  // schemas with `&` / `-`: the combine instantiations
        if (lds)
            hipLaunchKernelGGL((k_check_local<true, WAVES, true>), grid, dim3(WAVES * 64), prog_lds_bytes(g), s, g, items, n, rpw, nunits, nstatic, rdyn, next_unit, buf0, buf1, cap, overflow,
                               has, err, perm_out, err_out, max_level, skew, done_ctr, done_flag, done_val, inl);
        else
            hipLaunchKernelGGL((k_check_local<false, WAVES, true>), grid, dim3(WAVES * 64), 0, s, g, items, n, rpw, nunits, nstatic, rdyn, next_unit, buf0, buf1, cap, overflow, has, err,
                               perm_out, err_out, max_level, skew, done_ctr, done_flag, done_val, inl);
        return;
    }
    if (lds)
        hipLaunchKernelGGL((k_check_local<true, WAVES>), grid, dim3(WAVES * 64), prog_lds_bytes(g), s, g, items, n, rpw, nunits, nstatic, rdyn, next_unit, buf0, buf1, cap, overflow, has, err,
                           perm_out, err_out, max_level, skew, done_ctr, done_flag, done_val, inl);
    else
        hipLaunchKernelGGL((k_check_local<false, WAVES>), grid, dim3(WAVES * 64), 0, s, g, items, n, rpw, nunits, nstatic, rdyn, next_unit, buf0, buf1, cap, overflow, has, err, perm_out,
                           err_out, max_level, skew, done_ctr, done_flag, done_val, inl);
}
void launch_check_local(hipStream_t s, const DevGraph &g, const uint4 *items, uint32_t n, uint32_t rpw, uint32_t nblocks, uint32_t *next_unit, uint4 *buf0,
                        uint4 *buf1, uint32_t cap, uint32_t *overflow, uint8_t *has, uint8_t *err, uint8_t *perm_out, int32_t *err_out, uint32_t *max_level,
                        uint32_t nstatic, uint32_t rdyn, bool wide, uint32_t skew, uint32_t *done_ctr, uint32_t *done_flag, uint32_t done_val, const uint4 *inline_items_host) {
    InlineItems inl{};
    if (inline_items_host && n <= 4) {  // (host memory: copied into the launch's arguments)
        for (uint32_t i = 0; i < n; i++) inl.v[i] = inline_items_host[i];
        items = nullptr;
    }
    uint32_t nunits = (n + rpw - 1) / rpw;
    if (nstatic && rdyn && (uint64_t)nstatic * rpw < n) skew = 0;  // (static units only)
    skew = std::min(skew, std::min(rpw - 1u, (wide ? kLocalWide : kLocalNarrow) * 64u - rpw));  // the largest unit still fits the block: thread i seeds request first + i
    skew = nunits > 1 && nunits <= 4096 ? (skew << 8) / nunits : 0u;                             // (the kernel's fixed-point form: 1/256ths per unit; u (nunits - u) skew < 2^32)
    if (nstatic && rdyn && (uint64_t)nstatic * rpw < n) nunits = nstatic + (n - nstatic * rpw + rdyn - 1) / rdyn;  // static units, then small ones
    else nstatic = nunits, rdyn = rpw;
    if (wide) launch_check_local_w<kLocalWide>(s, g, items, n, rpw, nblocks, nunits, nstatic, rdyn, next_unit, buf0, buf1, cap, overflow, has, err, perm_out, err_out, max_level, skew, done_ctr, done_flag, done_val, inl);
    else launch_check_local_w<kLocalNarrow>(s, g, items, n, rpw, nblocks, nunits, nstatic, rdyn, next_unit, buf0, buf1, cap, overflow, has, err, perm_out, err_out, max_level, skew, done_ctr, done_flag, done_val, inl);
}
template <int WAVES>
static int local_occupancy(bool lds, size_t prog_bytes) {
    int occ = 0;
    const hipError_t oe = lds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_check_local<true, WAVES>, WAVES * 64, prog_bytes)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_check_local<false, WAVES>, WAVES * 64, 0);
    return (oe != hipSuccess || occ <= 0) ? std::max(1, 16 / WAVES) : occ;
}
int local_grid_blocks(int device, size_t prog_bytes, bool wide) {
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const bool lds = prog_in_lds() && prog_bytes <= (size_t)kProgLdsEntries * 32;
    return cus * (wide ? local_occupancy<kLocalWide>(lds, prog_bytes) : local_occupancy<kLocalNarrow>(lds, prog_bytes));
}
uint32_t local_unit_max(bool wide) { return (uint32_t)(wide ? kLocalWide : kLocalNarrow) * 64u; }
void launch_dedup(hipStream_t s, const DevFrontier &f, uint32_t iter, uint64_t *table, uint32_t bits, bool cells) {
    (void)hipMemsetAsync(table, 0xFF, sizeof(uint64_t) << bits, s);
    if (cells) {  // (`table` holds 2^bits keys and, behind them, 2^bits second halves)
        uint32_t *second = reinterpret_cast<uint32_t *>(table + ((size_t)1 << bits));
        (void)hipMemsetAsync(second, 0, sizeof(uint32_t) << bits, s);
        hipLaunchKernelGGL(k_dedup_cells, dim3(f.nwaves / kWavesPerBlock), dim3(256), 0, s, f, iter, reinterpret_cast<unsigned long long *>(table), second, bits);
        return;
    }
    hipLaunchKernelGGL(k_dedup, dim3(f.nwaves / kWavesPerBlock), dim3(256), 0, s, f, iter, reinterpret_cast<unsigned long long *>(table), bits);
}
void launch_finalize(hipStream_t s, uint32_t n, const uint8_t *has, const uint8_t *err, uint8_t *perm_out, int32_t *err_out) {
    if (!n) return;
    hipLaunchKernelGGL(k_finalize, dim3((n + 255) / 256), dim3(256), 0, s, n, has, err, perm_out, err_out);
}
void launch_resolve(hipStream_t s, const DevGraph &g, uint32_t iter, uint8_t *has, uint8_t *err) {
    hipLaunchKernelGGL(k_resolve, dim3(256), dim3(256), 0, s, g, iter, 1u, has, err);
    hipLaunchKernelGGL(k_resolve, dim3(256), dim3(256), 0, s, g, iter, 0u, has, err);
}
void launch_node_hdr(hipStream_t s, uint4 *hdr, const uint32_t *ccount) { hipLaunchKernelGGL(k_node_hdr, dim3(1), dim3(1), 0, s, hdr, ccount); }
void launch_resolve_gathered(hipStream_t s, const DevGraph &g, const uint4 *nodes, uint32_t stride, const uint4 *hdrs, uint32_t world, uint32_t iter, uint8_t *has, uint8_t *err) {
    hipLaunchKernelGGL(k_resolve_gathered, dim3(64, world), dim3(256), 0, s, g, nodes, stride, hdrs, iter, 1u, has, err);
    hipLaunchKernelGGL(k_resolve_gathered, dim3(64, world), dim3(256), 0, s, g, nodes, stride, hdrs, iter, 0u, has, err);
}
void launch_rev_seed(hipStream_t s, const DevFrontier &f, const uint32_t *d_sids, uint32_t n, uint32_t key) {
    const uint32_t threads = std::max(std::max(n, f.nwaves), kStatusWords);
    hipLaunchKernelGGL(k_rev_seed, dim3((threads + 255) / 256), dim3(256), 0, s, f, d_sids, n, key);
}
void launch_keep(hipStream_t s, uint32_t k_items, const uint32_t *item_off, const uint8_t *perm, uint8_t *keep_out) {
    if (!k_items) return;
    hipLaunchKernelGGL(k_keep, dim3((k_items + 255) / 256), dim3(256), 0, s, k_items, item_off, perm, keep_out);
}
void launch_rev_expand(hipStream_t s, const DevReverse &r, const DevFrontier &f, uint32_t iter, uint32_t phase, const DevShard &sh) {
    const dim3 grid(f.nwaves / kWavesPerBlock);
    if (phase == REV_VISIT) hipLaunchKernelGGL(k_rev_expand<REV_VISIT>, grid, dim3(kBlock), 0, s, r, f, iter, sh);
    else if (phase == REV_EXPAND) hipLaunchKernelGGL(k_rev_expand<REV_EXPAND>, grid, dim3(kBlock), 0, s, r, f, iter, sh);
    else hipLaunchKernelGGL(k_rev_expand<REV_FUSED>, grid, dim3(kBlock), 0, s, r, f, iter, sh);
}
void launch_rev_local(hipStream_t s, const DevReverse &r, const uint32_t *sids, uint32_t n, uint32_t key, uint32_t target_slot, void *buf0, void *buf1,
                      uint32_t cap, uint32_t *out_bitmaps, uint32_t out_stride, uint32_t copy_words, uint64_t *out_counts, uint32_t *status, uint32_t lds_row_words,
                      uint32_t *done_ctr, uint32_t *done_flag, uint32_t done_val, const RevBigRows *big, const RevUseful *useful_p) {
    if (!n) return;
    RevUseful useful;
    for (uint32_t k = 0; k < kRevUsefulWords; k++) useful.w[k] = useful_p ? useful_p->w[k] : 0xFFFFFFFFu;
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void *>(k_rev_local), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRevLdsRowBytes) == hipSuccess;
    if (lds_row_words * 4u > (big_lds ? kRevLdsRowBytes : 32768u)) lds_row_words = 0;  // rows stay in HBM
    RevDefer dfr;
    if (!lds_row_words && big && big->tasks)
        dfr = RevDefer{reinterpret_cast<uint2 *>(big->tasks), big->task_count, big->levels, big->task_cap, big->defer_min ? big->defer_min : 4096u, big->bytemap, big->bytemap_stride};
    hipLaunchKernelGGL(k_rev_local, dim3(n), dim3(kRevLocalThreads), (size_t)lds_row_words * 4, s, r, sids, key, target_slot, (uint2 *)buf0, (uint2 *)buf1, cap, out_bitmaps,
                       out_stride, copy_words, (unsigned long long *)out_counts, status, lds_row_words, done_ctr, done_flag, done_val, dfr, useful);
    if (!dfr.tasks) return;
    target_slot &= ~kRevTargetSink;
    // the two chip-wide launches behind it: the deferred rows' ids, then the rows themselves (copy out, count, clear)
    const uint32_t per = std::max(1u, std::min(256u, 2048u / n));  // blocks per lookup: the chip for a single lookup, ~8 blocks per CU in all for a batch
    hipLaunchKernelGGL(k_rev_terminal, dim3(per, n), dim3(256), 0, s, r, target_slot, dfr);
    const uint32_t slices = std::max(1u, std::min(per, (out_stride + 1023u) / 1024u));
    hipLaunchKernelGGL(k_rev_rows, dim3(slices, n), dim3(256), 0, s, r, target_slot, n, out_bitmaps, out_stride, copy_words, (unsigned long long *)out_counts,
                       (unsigned long long *)big->counts, dfr, done_ctr, done_flag, done_val);
}
static uint32_t import_blocks(uint32_t n) {  // ~one 1024-entry chunk of input per wave, at most 256 blocks
    const uint32_t b = (n + 4095) / 4096;
    return b < 1 ? 1 : (b > 256 ? 256 : b);
}
void launch_import(hipStream_t s, const DevGraph &g, const DevFrontier &f, uint32_t iter, const uint4 *in, uint32_t n, const DevShard &sh) {
    if (!n) return;
    hipLaunchKernelGGL(k_import<true>, dim3(import_blocks(n)), dim3(256), 0, s, f, iter, in, n, g.progs, (const RevProg *)nullptr, sh.rank);
}
void launch_xhdr(hipStream_t s, uint4 *hdr, uint32_t nblocks, const uint32_t *exp_count, const uint32_t *any_iter, const uint32_t *overflow) {
    hipLaunchKernelGGL(k_xhdr, dim3(1), dim3(64), 0, s, hdr, nblocks, exp_count, any_iter, overflow);
}
void launch_import_gathered(hipStream_t s, const DevGraph &g, const DevFrontier &f, uint32_t iter, const uint4 *hdrs, const uint4 *data, uint32_t world, uint32_t rank,
                            uint32_t cap, bool have_data, uint32_t *ctrl) {
    const uint32_t bx = have_data ? std::max<uint32_t>(1, std::min<uint32_t>(64, (cap + 4095) / 4096)) : 1u;
    hipLaunchKernelGGL(k_import_gathered<true>, dim3(bx, world), dim3(256), 0, s, f, iter, hdrs, data, world, rank, cap, have_data ? 1u : 0u, g.progs, (const RevProg *)nullptr, ctrl);
}
void launch_rev_import_gathered(hipStream_t s, const DevReverse &r, const DevFrontier &f, uint32_t iter, const uint4 *hdrs, const uint4 *data, uint32_t world, uint32_t rank,
                                uint32_t cap, bool have_data, uint32_t *ctrl) {
    const uint32_t bx = have_data ? std::max<uint32_t>(1, std::min<uint32_t>(64, (cap + 4095) / 4096)) : 1u;
    hipLaunchKernelGGL(k_import_gathered<false>, dim3(bx, world), dim3(256), 0, s, f, iter, hdrs, data, world, rank, cap, have_data ? 1u : 0u, (const SlotProg *)nullptr, r.rprogs, ctrl);
}
void launch_rev_import(hipStream_t s, const DevReverse &r, const DevFrontier &f, uint32_t iter, const uint4 *in, uint32_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_import<false>, dim3(import_blocks(n)), dim3(256), 0, s, f, iter, in, n, (const SlotProg *)nullptr, r.rprogs, 0u);
}

}  // namespace acl
