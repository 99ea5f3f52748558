// kernels.hip -- gfx950 (CDNA4) frontier-expansion kernels of the batched ACL-check engine.
//
// What they replace: the per-item recursive dispatch SpiceDB runs for
// CheckBulkPermissions (reached from reference pkg/authz/check.go:48 and
// pkg/authz/postfilter.go:134) and the reverse walk behind LookupResources
// (pkg/authz/lookups.go:65).  Here a whole request batch advances together, one
// dispatch level per launch:
//
//   frontier entry = (request, object#relation state)                16 B
//   k_expand: every lane takes one entry, interprets the state's flattened
//             program (plan.hpp): binary-search probes of the sorted CSR row for the
//             request's subject, and "enumerate" operations whose child states are
//             produced by a wave-cooperative, load-balanced expansion:
//               - per-lane tasks (row start, degree) are compacted into LDS with
//                 wave64 ballot + mbcnt,
//               - a wave prefix-sum over task degrees sizes the output,
//               - lanes then take consecutive OUTPUT slots, find their task by
//                 binary search in the LDS prefix array, and load consecutive edges
//                 of a row (coalesced) and store consecutive 16 B entries (coalesced).
//   Output space comes from wave-private 16 KiB chunks, so the only global atomic
//   is one per ~1024 produced entries.
//
// Bound: HBM/L2 transactions (random row gathers); no MFMA anywhere by design.
#include "kernels.hpp"

namespace acl {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * 64;
constexpr uint32_t kTaskCap = 192;  // LDS task slots per wave
constexpr uint32_t kSelfBit = 0x80000000u;
constexpr uint32_t kNoChunk = 0xFFFFFFFFu;
constexpr uint32_t kMaxRow = 1u << 25;  // rows longer than this cannot be enumerated in one task

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d, 64));
    return v;
}

// LDS-staged task list of one wave + its private output chunk.
struct TaskLds {
    uint32_t start[kTaskCap];  // first edge (absolute index) -- or the object id for a "self" task
    uint32_t count[kTaskCap];  // degree (| kSelfBit)
    uint32_t req[kTaskCap];
    uint32_t meta[kTaskCap];   // child meta
    uint32_t sid[kTaskCap];
    uint32_t scan[64];
};

struct WaveOut {
    uint32_t cur = kNoChunk;  // wave-uniform
    uint32_t fill = 0;
};

// Reserve `total` consecutive output entries for this wave. Returns the first entry index
// or 0xFFFFFFFF when the frontier is out of chunks (overflow flag raised).
__device__ __forceinline__ uint32_t reserve(WaveOut &wo, uint32_t total, uint32_t lane, uint32_t *out_counts, uint32_t *out_nchunks,
                                            uint32_t max_chunks, uint32_t *overflow) {
    if (wo.cur != kNoChunk && wo.fill + total <= kChunk) {
        uint32_t base = wo.cur * kChunk + wo.fill;
        wo.fill += total;
        return base;
    }
    if (wo.cur != kNoChunk && lane == 0) out_counts[wo.cur] = wo.fill;
    uint32_t m = (total + kChunk - 1) / kChunk;
    uint32_t cb = 0;
    if (lane == 0) cb = atomicAdd(out_nchunks, m);
    cb = uniform(cb);
    if (cb + m > max_chunks || cb + m < cb) {
        if (lane == 0) *overflow = 1u;
        wo.cur = kNoChunk;
        wo.fill = 0;
        return 0xFFFFFFFFu;
    }
    for (uint32_t c = lane; c + 1 < m; c += 64) out_counts[cb + c] = kChunk;
    wo.cur = cb + m - 1;
    wo.fill = total - (m - 1) * kChunk;
    return cb * kChunk;
}

// Expand the first T tasks of the wave's LDS list into output entries.
__device__ __forceinline__ void flush_tasks(TaskLds &t, uint32_t T, WaveOut &wo, uint32_t lane, const uint32_t *__restrict__ edges,
                                            uint4 *__restrict__ out, uint32_t *out_counts, uint32_t *out_nchunks, uint32_t max_chunks,
                                            uint32_t *overflow) {
    wave_lds_fence();
    for (uint32_t g = 0; g < T; g += 64) {
        uint32_t cnt = (g + lane < T) ? (t.count[g + lane] & ~kSelfBit) : 0u;
        uint32_t incl = wave_incl_scan(cnt, lane);
        uint32_t total = uniform(__shfl(incl, 63, 64));
        t.scan[lane] = incl - cnt;
        wave_lds_fence();
        if (total) {
            uint32_t base = reserve(wo, total, lane, out_counts, out_nchunks, max_chunks, overflow);
            if (base != 0xFFFFFFFFu) {
                for (uint32_t w0 = 0; w0 < total; w0 += 64) {
                    uint32_t w = w0 + lane;
                    if (w < total) {
                        // largest j with scan[j] <= w
                        uint32_t j = 0;
#pragma unroll
                        for (uint32_t step = 32; step >= 1; step >>= 1)
                            if (t.scan[j + step] <= w) j += step;
                        uint32_t tj = g + j;
                        uint32_t c = t.count[tj], s = t.start[tj];
                        uint32_t child = (c & kSelfBit) ? s : edges[s + (w - t.scan[j])];
                        out[base + w] = make_uint4(child, t.req[tj], t.meta[tj], t.sid[tj]);
                    }
                }
            }
        }
        wave_lds_fence();
    }
}

__device__ __forceinline__ bool row_contains(const uint32_t *__restrict__ edges, uint32_t lo, uint32_t hi, uint32_t key) {
    uint32_t end = hi;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (edges[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo < end && edges[lo] == key;
}

// ------------------------------------------------------------------ seed
// items: acl_item_t (16 B): x = rtype | perm << 16, y = resource id, z = stype | srel << 16, w = subject id
__global__ __launch_bounds__(256) void k_seed(DevGraph g, DevFrontier f, const uint4 *__restrict__ items, uint32_t n, uint8_t *has, uint8_t *err) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) f.nchunks[0] = (n + kChunk - 1) / kChunk;
    if (i < (n + kChunk - 1) / kChunk) f.counts[0][i] = min(kChunk, n - i * kChunk);
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
        meta = slot | (1u << 13) | (key << 19);
    }
    f.buf[0][i] = make_uint4(it.y, i, meta, it.w);
}

// ---------------------------------------------------------------- expand
__global__ __launch_bounds__(kBlock) void k_expand(DevGraph g, DevFrontier f, uint32_t iter, uint8_t *has, uint8_t *err) {
    __shared__ TaskLds lds[kWavesPerBlock];
    const uint32_t lane = lane_id();
    const uint32_t wib = threadIdx.x >> 6;
    TaskLds &t = lds[wib];
    const uint32_t wave = blockIdx.x * kWavesPerBlock + wib, nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t pin = (iter + 1) & 1u, pout = iter & 1u;  // iteration i reads parity (i-1)&1
    const uint4 *__restrict__ in = f.buf[pin];
    const uint32_t *__restrict__ in_counts = f.counts[pin];
    uint4 *__restrict__ out = f.buf[pout];
    uint32_t *out_counts = f.counts[pout];
    uint32_t *out_nchunks = f.nchunks + iter;
    if (*f.overflow) return;
    uint32_t C = min(f.nchunks[iter - 1], f.max_chunks);
    WaveOut wo;
    for (uint32_t x = wave; x < C * kSegsPerChunk; x += nwaves) {
        const uint32_t c = x / kSegsPerChunk, s = x % kSegsPerChunk;
        const uint32_t cnt = in_counts[c];
        if (s * 64 >= cnt) continue;
        const bool valid = s * 64 + lane < cnt;
        uint4 e = valid ? in[(size_t)c * kChunk + s * 64 + lane] : make_uint4(0, 0, kDeadMeta, 0);
        const uint32_t id = e.x, req = e.y, meta = e.z, sid = e.w;
        bool active = valid && meta != kDeadMeta;
        if (active && has[req]) active = false;  // request already answered HAS: drop its pending work
        const uint32_t slot = meta & 0x1FFFu, level = (meta >> 13) & 63u, key = meta >> 19;
        SlotProg p = active ? g.progs[slot] : SlotProg{0, 0, 0, 0};
        const uint32_t nops = active ? (key < g.nslots ? p.n_total : p.n_main) : 0u;
        bool depth_err = active && level + p.max_dlevel > kMaxLevels;
        bool hit = false;
        uint32_t T = 0;
        const uint32_t maxops = uniform(wave_max(nops));
        for (uint32_t j = 0; j < maxops; j++) {
            bool want = false;
            uint32_t tstart = 0, tcount = 0, tmeta = 0;
            if (j < nops) {
                const FwdOp op = g.ops[p.first + j];
                const uint32_t L = level + op.dlevel;
                if (L <= kMaxLevels) {
                    if (op.flags & OP_REFLEX) {
                        if (key == op.key && id == sid) hit = true;
                    } else if (op.flags & OP_PUSH_SAME) {
                        if (L + 1 > kMaxLevels) depth_err = true;
                        else {
                            want = true;
                            tstart = id;
                            tcount = 1u | kSelfBit;
                            tmeta = op.key | ((L + 1) << 13) | (key << 19);
                        }
                    } else if (id < op.nrows) {
                        const uint32_t *o = g.off + op.off_base + (size_t)id * op.K + op.k;
                        const uint32_t s0 = o[0], s1 = o[1];
                        if (s1 > s0) {
                            if ((op.flags & OP_PROBE) && key == op.key && row_contains(g.edges, s0, s1, sid)) hit = true;
                            if (op.flags & OP_ENUM) {
                                if (L + 1 > kMaxLevels) depth_err = true;
                                else if (s1 - s0 > kMaxRow) *f.overflow = 2u;
                                else {
                                    want = true;
                                    tstart = s0;
                                    tcount = s1 - s0;
                                    tmeta = (op.flags & OP_PROBE ? op.key : op.key) | ((L + 1) << 13) | (key << 19);
                                }
                            }
                        }
                    }
                }
            }
            const uint64_t b = __ballot(want);
            if (b) {
                if (want) {
                    const uint32_t q = T + lanes_below(b);
                    t.start[q] = tstart;
                    t.count[q] = tcount;
                    t.req[q] = req;
                    t.meta[q] = tmeta;
                    t.sid[q] = sid;
                }
                T += (uint32_t)__popcll(b);
                if (T > kTaskCap - 64) {
                    flush_tasks(t, T, wo, lane, g.edges, out, out_counts, out_nchunks, f.max_chunks, f.overflow);
                    T = 0;
                }
            }
        }
        if (hit) has[req] = 1;
        else if (depth_err) err[req] = ITEM_ERR_DEPTH;
        if (T) flush_tasks(t, T, wo, lane, g.edges, out, out_counts, out_nchunks, f.max_chunks, f.overflow);
    }
    if (wo.cur != kNoChunk && lane == 0) out_counts[wo.cur] = wo.fill;
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

// ----------------------------------------------------------- reverse expand
// entry: x = object id, y = lookup request, z = meta (slot | dist << 13), w unused.
// dist == 0 marks a seed entry: slot field holds the SUBJECT KEY and the program is rseeds[key].
__global__ __launch_bounds__(kBlock) void k_rev_expand(DevReverse r, DevFrontier f, uint32_t iter, uint32_t nslots) {
    __shared__ TaskLds lds[kWavesPerBlock];
    const uint32_t lane = lane_id();
    const uint32_t wib = threadIdx.x >> 6;
    TaskLds &t = lds[wib];
    const uint32_t wave = blockIdx.x * kWavesPerBlock + wib, nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t pin = (iter + 1) & 1u, pout = iter & 1u;
    const uint4 *__restrict__ in = f.buf[pin];
    const uint32_t *__restrict__ in_counts = f.counts[pin];
    uint4 *__restrict__ out = f.buf[pout];
    uint32_t *out_counts = f.counts[pout];
    uint32_t *out_nchunks = f.nchunks + iter;
    if (*f.overflow) return;
    uint32_t C = min(f.nchunks[iter - 1], f.max_chunks);
    WaveOut wo;
    for (uint32_t x = wave; x < C * kSegsPerChunk; x += nwaves) {
        const uint32_t c = x / kSegsPerChunk, s = x % kSegsPerChunk;
        const uint32_t cnt = in_counts[c];
        if (s * 64 >= cnt) continue;
        const bool valid = s * 64 + lane < cnt;
        uint4 e = valid ? in[(size_t)c * kChunk + s * 64 + lane] : make_uint4(0, 0, kDeadMeta, 0);
        const uint32_t id = e.x, req = e.y, meta = e.z;
        bool active = valid && meta != kDeadMeta;
        const uint32_t slot = meta & 0x1FFFu, dist = (meta >> 13) & 63u;
        RevProg p{0, 0};
        if (active) {
            if (dist == 0) {
                p = r.rseeds[slot];
            } else if (id < r.slot_nobjects[slot]) {
                // first visit wins: level-synchronous order makes it the minimum distance
                const uint32_t bit = r.slot_bit_base[slot] + id;
                const uint32_t m = 1u << (bit & 31u);
                const uint32_t old = atomicOr(r.visited + (size_t)req * r.visited_words + (bit >> 5), m);
                if (old & m) active = false;
                else p = r.rprogs[slot];
            } else {
                active = false;
            }
        }
        const uint32_t nops = (active && dist < kMaxLevels) ? p.n : 0u;  // parents of a dist-50 state would need 51 levels
        uint32_t T = 0;
        const uint32_t maxops = uniform(wave_max(nops));
        for (uint32_t j = 0; j < maxops; j++) {
            bool want = false;
            uint32_t tstart = 0, tcount = 0, tmeta = 0;
            if (j < nops) {
                const RevOp op = r.rops[p.first + j];
                if (op.flags & OP_PUSH_SAME) {
                    want = true;
                    tstart = id;
                    tcount = 1u | kSelfBit;
                } else if (id < op.nrows) {
                    const uint32_t s0 = r.roff[op.roff_base + id], s1 = r.roff[op.roff_base + id + 1];
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
                    t.start[q] = tstart;
                    t.count[q] = tcount;
                    t.req[q] = req;
                    t.meta[q] = tmeta;
                    t.sid[q] = 0;
                }
                T += (uint32_t)__popcll(b);
                if (T > kTaskCap - 64) {
                    flush_tasks(t, T, wo, lane, r.redges, out, out_counts, out_nchunks, f.max_chunks, f.overflow);
                    T = 0;
                }
            }
        }
        if (T) flush_tasks(t, T, wo, lane, r.redges, out, out_counts, out_nchunks, f.max_chunks, f.overflow);
    }
    if (wo.cur != kNoChunk && lane == 0) out_counts[wo.cur] = wo.fill;
    (void)nslots;
}

}  // namespace

void launch_seed(hipStream_t s, const DevGraph &g, const DevFrontier &f, const uint4 *items, uint32_t n, uint32_t, uint8_t *has, uint8_t *err) {
    if (!n) return;
    hipLaunchKernelGGL(k_seed, dim3((n + 255) / 256), dim3(256), 0, s, g, f, items, n, has, err);
}
void launch_expand(hipStream_t s, int grid_blocks, const DevGraph &g, const DevFrontier &f, uint32_t iter, uint8_t *has, uint8_t *err) {
    hipLaunchKernelGGL(k_expand, dim3(grid_blocks), dim3(kBlock), 0, s, g, f, iter, has, err);
}
void launch_finalize(hipStream_t s, uint32_t n, const uint8_t *has, const uint8_t *err, uint8_t *perm_out, int32_t *err_out) {
    if (!n) return;
    hipLaunchKernelGGL(k_finalize, dim3((n + 255) / 256), dim3(256), 0, s, n, has, err, perm_out, err_out);
}
void launch_rev_expand(hipStream_t s, int grid_blocks, const DevReverse &r, const DevFrontier &f, uint32_t iter, uint32_t nslots) {
    hipLaunchKernelGGL(k_rev_expand, dim3(grid_blocks), dim3(kBlock), 0, s, r, f, iter, nslots);
}

}  // namespace acl
