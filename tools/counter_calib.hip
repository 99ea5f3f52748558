// counter_calib.hip -- known-byte-count micro-kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS engine's access
// patterns (VERDICT r3 next #1a): the guide (MI355X_MICROARCH.md, HBM section) calibrates only a wide coalesced read stream (counter = 1/2
// of the bytes) and calls every other width "uncalibrated"; k_check_local is 16 B bucket gathers, 8 B descriptor gathers, 4 B edge
// gathers and compacted 16 B entry stores.  Every kernel here moves an exactly known number of bytes in one of those shapes:
//   stream_read16   coalesced 16 B per lane over the whole array (the guide's calibrated case: expect counter = bytes / 2)
//   gather4/8/16    one random, naturally aligned 4 / 8 / 16 B load per lane
//   gather64        one random 64 B-aligned segment per lane, read as 4 x 16 B
//   gather16x4      4 neighbouring lanes share one random 64 B segment (16 B each): the shape of a wave probing one subject's buckets
//   stream_write16  coalesced 16 B per lane
//   scatter8/16     one random, naturally aligned 8 / 16 B store per lane
//   append16        wave-compacted consecutive 16 B stores behind a per-wave cursor (the frontier's writes)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again with `--pmc WRITE_SIZE` (tools/runs/gpu_r04_1.sh);
// tools/calib_summary.py turns the two databases into profiles/r04_counter_calibration.md.
// Build: hipcc --offload-arch=gfx950 -O3 tools/counter_calib.hip -o tools/bin/counter_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64 finaliser: a different address per (launch seed, work item)
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather(const T *__restrict__ a, uint64_t nelem, uint64_t nwork, uint64_t seed, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwork; i += (uint64_t)gridDim.x * blockDim.x) {
        const T v = a[mix(i ^ seed) % nelem];
        acc ^= *reinterpret_cast<const uint32_t *>(&v);
    }
    if (acc == 0x12345678u) *sink = acc;
}
// G lanes share one random 64 B segment; each reads 64 / G bytes of it as 16 B pieces
template <int G>
__global__ __launch_bounds__(256) void k_gather_seg64(const uint4 *__restrict__ a, uint64_t nseg, uint64_t nwork, uint64_t seed, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwork; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t seg = mix((i / G) ^ seed) % nseg;
#pragma unroll
        for (int k = 0; k < 4 / G; k++) {
            const uint4 v = a[seg * 4 + (i % G) * (4 / G) + k];
            acc ^= v.x ^ v.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_stream_read16(const uint4 *__restrict__ a, uint64_t nelem, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nelem; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = a[i];
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_stream_write16(uint4 *a, uint64_t nelem, uint32_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nelem; i += (uint64_t)gridDim.x * blockDim.x)
        a[i] = make_uint4((uint32_t)i, seed, 0u, 1u);
}
template <typename T>
__global__ __launch_bounds__(256) void k_scatter(T *a, uint64_t nelem, uint64_t nwork, uint64_t seed) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwork; i += (uint64_t)gridDim.x * blockDim.x) {
        T v;
        memset(&v, 0, sizeof(T));
        *reinterpret_cast<uint32_t *>(&v) = (uint32_t)i;
        a[mix(i ^ seed) % nelem] = v;
    }
}
// every wave appends ~half of its lanes' entries (ballot-compacted, consecutive) behind its own cursor: region = nelem / waves entries
template <typename T>
__global__ __launch_bounds__(256) void k_append(T *a, uint64_t per_wave, uint64_t rounds, uint64_t seed, unsigned long long *written) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    T *out = a + wave * per_wave;
    uint64_t cur = 0;
    for (uint64_t r = 0; r < rounds; r++) {
        const bool push = (mix((wave * rounds + r) * 64 + lane + seed) & 1ull) != 0;
        const uint64_t b = __ballot(push);
        const uint32_t n = (uint32_t)__popcll(b);
        if (cur + n > per_wave) break;
        if (push) {
            T v;
            memset(&v, 0, sizeof(T));
            *reinterpret_cast<uint32_t *>(&v) = (uint32_t)r;
            out[cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = v;
        }
        cur += n;
    }
    if (lane == 0) atomicAdd(written, (unsigned long long)cur);
}

int main(int argc, char **argv) {
    const uint64_t big = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 30);     // bytes of the large table (beyond the Infinity Cache)
    const uint64_t small = argc > 2 ? strtoull(argv[2], nullptr, 0) : (64ull << 20);   // bytes of the small table (C4's snapshot is 73 MB)
    const uint64_t work = argc > 3 ? strtoull(argv[3], nullptr, 0) : (1ull << 26);     // accesses per gather / scatter launch
    const int reps = 3;
    uint4 *a;
    uint32_t *sink;
    unsigned long long *written;
    CK(hipMalloc(&a, big));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&written, 8));
    CK(hipMemset(a, 0x5A, big));
    CK(hipMemset(sink, 0, 4));
    const dim3 grid(256 * 16), block(256);
    // one line per launch: name, table bytes, algorithmic read bytes, algorithmic write bytes (the summary joins on launch order)
    auto say = [](const char *name, uint64_t table, uint64_t rd, uint64_t wr) { printf("CALIB %s %llu %llu %llu\n", name, (unsigned long long)table, (unsigned long long)rd, (unsigned long long)wr); };
    for (int rep = 0; rep < reps; rep++) {
        const uint64_t seed = 0x5ACE0000ull + (uint64_t)rep * 7919ull;
        for (uint64_t tb : {big, small}) {
            hipLaunchKernelGGL(k_stream_read16, grid, block, 0, 0, a, tb / 16, sink);
            say("stream_read16", tb, tb, 0);
            hipLaunchKernelGGL(k_gather<uint32_t>, grid, block, 0, 0, reinterpret_cast<const uint32_t *>(a), tb / 4, work, seed, sink);
            say("gather4", tb, work * 4, 0);
            hipLaunchKernelGGL(k_gather<uint2>, grid, block, 0, 0, reinterpret_cast<const uint2 *>(a), tb / 8, work, seed, sink);
            say("gather8", tb, work * 8, 0);
            hipLaunchKernelGGL(k_gather<uint4>, grid, block, 0, 0, a, tb / 16, work, seed, sink);
            say("gather16", tb, work * 16, 0);
            hipLaunchKernelGGL(k_gather_seg64<1>, grid, block, 0, 0, a, tb / 64, work / 4, seed, sink);
            say("gather64", tb, work / 4 * 64, 0);
            hipLaunchKernelGGL(k_gather_seg64<4>, grid, block, 0, 0, a, tb / 64, work, seed, sink);
            say("gather16x4", tb, work * 16, 0);
        }
        for (uint64_t tb : {big, small}) {
            hipLaunchKernelGGL(k_stream_write16, grid, block, 0, 0, a, tb / 16, (uint32_t)seed);
            say("stream_write16", tb, 0, tb);
            hipLaunchKernelGGL(k_scatter<uint2>, grid, block, 0, 0, reinterpret_cast<uint2 *>(a), tb / 8, work, seed);
            say("scatter8", tb, 0, work * 8);
            hipLaunchKernelGGL(k_scatter<uint4>, grid, block, 0, 0, a, tb / 16, work, seed);
            say("scatter16", tb, 0, work * 16);
        }
        {
            const uint64_t waves = (uint64_t)grid.x * block.x / 64;
            unsigned long long w16 = 0, w8 = 0;
            CK(hipMemset(written, 0, 8));
            hipLaunchKernelGGL(k_append<uint4>, grid, block, 0, 0, a, big / 16 / waves, (big / 16 / waves) / 32, seed, written);
            CK(hipMemcpy(&w16, written, 8, hipMemcpyDeviceToHost));
            say("append16", big, 0, w16 * 16);
            CK(hipMemset(written, 0, 8));
            hipLaunchKernelGGL(k_append<uint2>, grid, block, 0, 0, reinterpret_cast<uint2 *>(a), big / 8 / waves, (big / 8 / waves) / 32, seed, written);
            CK(hipMemcpy(&w8, written, 8, hipMemcpyDeviceToHost));
            say("append8", big, 0, w8 * 8);
        }
        CK(hipDeviceSynchronize());
    }
    CK(hipFree(a));
    return 0;
}
