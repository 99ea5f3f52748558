// ipc_comm.hip -- an acl_shard_comm_t (include/aclgpu.h) between PROCESSES that share one GPU (or sit on GPUs of one node), built on
// hipIpcGetMemHandle / hipIpcOpenMemHandle and a barrier in POSIX shared memory.  -> tools/bin/libaclipc.so (built by __graft_entry__.build()
// and by tests/test_sharded_gpu.py on demand).
//
// Why it exists: the sharded level loops of engine_shard_native.cpp had only ever run between logical shards of ONE
// process -- threads, one HIP context, every buffer addressable by everybody -- or over RCCL with world 1; RCCL refuses two ranks on one device,
// and the test boxes have one.  The callbacks interface does not care who moves the bytes: this communicator moves them between two address
// spaces, two HIP contexts and two sets of streams on the same device, which is what the in-process double cannot stand in for.  TEST
// INFRASTRUCTURE: production uses acl_shard_rccl_* (RCCL over xGMI); nothing in libaclgpu.so links or loads this file.
//
// Mechanics.  Every rank owns one WINDOW of device memory, exported once through the shared segment and mapped by every peer at open.  A
// collective stages the caller's send buffer into the own window (device-to-device copy on the caller's stream), synchronises that stream,
// meets the peers at the barrier, pulls what it needs out of the peers' windows (copies -- or a byte-wise max kernel -- on the caller's stream
// again), synchronises, and meets the peers once more so that nobody overwrites a window somebody is still reading.  Transfers larger than the
// window go in slices.  A peer that dies or hangs cannot wedge the others: every barrier wait has a deadline, and the first rank to give up
// poisons the segment so the rest fail at once.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "aclgpu.h"

namespace {

constexpr uint32_t kMaxWorld = 16;
constexpr uint32_t kMagic = 0xAC11C0DEu;

struct Segment {
    std::atomic<uint32_t> magic;       // set by rank 0 once the segment is zeroed
    std::atomic<uint32_t> arrived;     // barrier: ranks that reached the current generation
    std::atomic<uint32_t> generation;  // barrier: bumped by the last arrival
    std::atomic<uint32_t> poisoned;    // somebody gave up (deadline, HIP error): everybody fails from here on
    std::atomic<uint32_t> ready[kMaxWorld];
    hipIpcMemHandle_t window[kMaxWorld];
    uint64_t window_bytes[kMaxWorld];
};

thread_local std::string g_err;

int64_t mono_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

__global__ void k_max_u8(uint8_t *dst, const uint8_t *src, size_t n) {
    // 4 bytes per thread where both sides allow it (windows and slice offsets are 256-byte aligned; the caller's buffer usually is)
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n && (((uintptr_t)dst | (uintptr_t)src) & 3u) == 0) {
        const uint32_t a = *(const uint32_t *)(dst + i), b = *(const uint32_t *)(src + i);
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 32; k += 8) {
            const uint32_t x = (a >> k) & 0xFFu, y = (b >> k) & 0xFFu;
            r |= (x > y ? x : y) << k;
        }
        *(uint32_t *)(dst + i) = r;
        return;
    }
    for (size_t k = i; k < n && k < i + 4; k++) dst[k] = dst[k] > src[k] ? dst[k] : src[k];
}

}  // namespace

struct aclipc {
    std::string name;
    uint32_t rank = 0, world = 1;
    int device = 0;
    size_t window_bytes = 0;
    Segment *seg = nullptr;
    uint8_t *mine = nullptr;
    std::vector<uint8_t *> win;  // [world]: peers' windows as mapped here; win[rank] == mine
    int64_t deadline_ns = 60ll * 1000000000ll;
    // what moved: [0] collectives, [1] bytes read out of FOREIGN windows, [2] barriers, [3] longest barrier wait (ns)
    uint64_t stat[4] = {0, 0, 0, 0};

    int fail(const std::string &m) {
        g_err = "aclipc rank " + std::to_string(rank) + ": " + m;
        if (seg) seg->poisoned.store(1);
        return ACL_ERR_INTERNAL;
    }
    int hip(hipError_t e, const char *what) { return e == hipSuccess ? 0 : fail(std::string(what) + ": " + hipGetErrorString(e)); }

    int barrier() {
        const int64_t t0 = mono_ns();
        const uint32_t gen = seg->generation.load(std::memory_order_acquire);
        if (seg->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == world) {
            seg->arrived.store(0, std::memory_order_relaxed);
            seg->generation.fetch_add(1, std::memory_order_release);
        } else {
            for (uint32_t spins = 0; seg->generation.load(std::memory_order_acquire) == gen; spins++) {
                if (seg->poisoned.load(std::memory_order_relaxed)) return fail("a peer failed (segment poisoned)");
                if (spins > 2000) {
                    sched_yield();
                    if ((spins & 1023u) == 0 && mono_ns() - t0 > deadline_ns) return fail("barrier deadline exceeded: a peer died or hangs");
                }
            }
        }
        if (seg->poisoned.load(std::memory_order_relaxed)) return fail("a peer failed (segment poisoned)");
        stat[2]++;
        const uint64_t w = (uint64_t)(mono_ns() - t0);
        if (w > stat[3]) stat[3] = w;
        return 0;
    }

    int all_gather(const void *send, void *recv, size_t bytes, hipStream_t s) {
        stat[0]++;
        for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += window_bytes) {
            const size_t len = bytes - off < window_bytes ? bytes - off : window_bytes;
            if (len && hip(hipMemcpyAsync(mine, (const uint8_t *)send + off, len, hipMemcpyDeviceToDevice, s), "stage")) return ACL_ERR_INTERNAL;
            if (hip(hipStreamSynchronize(s), "sync (staged)")) return ACL_ERR_INTERNAL;
            if (barrier()) return ACL_ERR_INTERNAL;
            for (uint32_t r = 0; r < world && len; r++) {
                if (hip(hipMemcpyAsync((uint8_t *)recv + (size_t)r * bytes + off, win[r], len, hipMemcpyDeviceToDevice, s), "pull")) return ACL_ERR_INTERNAL;
                if (r != rank) stat[1] += len;
            }
            if (hip(hipStreamSynchronize(s), "sync (pulled)")) return ACL_ERR_INTERNAL;
            if (barrier()) return ACL_ERR_INTERNAL;
            if (bytes == 0) break;
        }
        return ACL_OK;
    }

    // block r of `send` -> rank r's block `rank` of `recv`: the window is cut into `world` lanes, one per destination
    int all_to_all(const void *send, void *recv, size_t bytes, hipStream_t s) {
        stat[0]++;
        const size_t lane = (window_bytes / world) & ~(size_t)255;
        if (!lane) return fail("window too small for an all-to-all");
        for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += lane) {
            const size_t len = bytes - off < lane ? bytes - off : lane;
            for (uint32_t r = 0; r < world && len; r++)
                if (hip(hipMemcpyAsync(mine + (size_t)r * lane, (const uint8_t *)send + (size_t)r * bytes + off, len, hipMemcpyDeviceToDevice, s), "stage")) return ACL_ERR_INTERNAL;
            if (hip(hipStreamSynchronize(s), "sync (staged)")) return ACL_ERR_INTERNAL;
            if (barrier()) return ACL_ERR_INTERNAL;
            for (uint32_t r = 0; r < world && len; r++) {
                if (hip(hipMemcpyAsync((uint8_t *)recv + (size_t)r * bytes + off, win[r] + (size_t)rank * lane, len, hipMemcpyDeviceToDevice, s), "pull")) return ACL_ERR_INTERNAL;
                if (r != rank) stat[1] += len;
            }
            if (hip(hipStreamSynchronize(s), "sync (pulled)")) return ACL_ERR_INTERNAL;
            if (barrier()) return ACL_ERR_INTERNAL;
            if (bytes == 0) break;
        }
        return ACL_OK;
    }

    int all_reduce_max(void *buf, size_t n, hipStream_t s) {
        stat[0]++;
        for (size_t off = 0; off < n || (n == 0 && off == 0); off += window_bytes) {
            const size_t len = n - off < window_bytes ? n - off : window_bytes;
            if (len && hip(hipMemcpyAsync(mine, (uint8_t *)buf + off, len, hipMemcpyDeviceToDevice, s), "stage")) return ACL_ERR_INTERNAL;
            if (hip(hipStreamSynchronize(s), "sync (staged)")) return ACL_ERR_INTERNAL;
            if (barrier()) return ACL_ERR_INTERNAL;
            for (uint32_t r = 0; r < world && len; r++) {
                if (r == rank) continue;
                const size_t threads = (len + 3) / 4;
                k_max_u8<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>((uint8_t *)buf + off, win[r], len);
                if (hip(hipGetLastError(), "max kernel")) return ACL_ERR_INTERNAL;
                stat[1] += len;
            }
            if (hip(hipStreamSynchronize(s), "sync (reduced)")) return ACL_ERR_INTERNAL;
            if (barrier()) return ACL_ERR_INTERNAL;
            if (n == 0) break;
        }
        return ACL_OK;
    }
};

namespace {
int cb_all_gather(void *u, const void *s, void *r, size_t b, void *st) { return ((aclipc *)u)->all_gather(s, r, b, (hipStream_t)st); }
int cb_all_reduce(void *u, void *b, size_t n, void *st) { return ((aclipc *)u)->all_reduce_max(b, n, (hipStream_t)st); }
int cb_all_to_all(void *u, const void *s, void *r, size_t b, void *st) { return ((aclipc *)u)->all_to_all(s, r, b, (hipStream_t)st); }
}  // namespace

extern "C" {

const char *aclipc_last_error() { return g_err.c_str(); }

// shm_name: a POSIX shared-memory name ("/aclipc-<something unique per test>"); rank 0 creates the segment, the others wait for it.
// deadline_s: how long a barrier (and the rendezvous at open) waits for the peers before the whole communicator fails.
int aclipc_open(const char *shm_name, uint32_t rank, uint32_t world, int device, size_t window_bytes, int deadline_s, aclipc **out) {
    if (!shm_name || !out || world == 0 || world > kMaxWorld || rank >= world) {
        g_err = "aclipc_open: bad arguments";
        return ACL_ERR_INVALID_ARGUMENT;
    }
    auto *c = new aclipc();
    c->name = shm_name;
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->window_bytes = (window_bytes < 4096 ? 4096 : window_bytes) & ~(size_t)255;
    if (deadline_s > 0) c->deadline_ns = (int64_t)deadline_s * 1000000000ll;
    const int64_t t0 = mono_ns();
    int fd = -1;
    if (rank == 0) {
        shm_unlink(shm_name);  // (a stale segment of a killed run)
        fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) {
            c->fail(std::string("shm_open / ftruncate: ") + strerror(errno));
            delete c;
            return ACL_ERR_INTERNAL;
        }
    } else {
        for (;;) {
            fd = shm_open(shm_name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(Segment)) break;
            if (fd >= 0) close(fd);
            if (mono_ns() - t0 > c->deadline_ns) {
                c->fail("rank 0 never created the segment");
                delete c;
                return ACL_ERR_INTERNAL;
            }
            usleep(2000);
        }
    }
    void *m = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        c->fail(std::string("mmap: ") + strerror(errno));
        delete c;
        return ACL_ERR_INTERNAL;
    }
    c->seg = (Segment *)m;
    if (rank == 0) {
        std::memset(m, 0, sizeof(Segment));
        c->seg->magic.store(kMagic, std::memory_order_release);
    } else {
        while (c->seg->magic.load(std::memory_order_acquire) != kMagic) {
            if (mono_ns() - t0 > c->deadline_ns) {
                c->fail("segment never initialised");
                delete c;
                return ACL_ERR_INTERNAL;
            }
            usleep(1000);
        }
    }
    auto bail = [&](int rc) {
        *out = nullptr;
        if (c->mine) (void)hipFree(c->mine);
        munmap(c->seg, sizeof(Segment));
        delete c;
        return rc;
    };
    if (c->hip(hipSetDevice(device), "hipSetDevice") || c->hip(hipMalloc((void **)&c->mine, c->window_bytes), "hipMalloc (window)") ||
        c->hip(hipMemset(c->mine, 0, c->window_bytes), "hipMemset (window)") ||
        c->hip(hipIpcGetMemHandle(&c->seg->window[rank], c->mine), "hipIpcGetMemHandle (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"))
        return bail(ACL_ERR_INTERNAL);
    c->seg->window_bytes[rank] = c->window_bytes;
    c->seg->ready[rank].store(1, std::memory_order_release);
    c->win.assign(world, nullptr);
    c->win[rank] = c->mine;
    for (uint32_t r = 0; r < world; r++) {
        if (r == rank) continue;
        while (!c->seg->ready[r].load(std::memory_order_acquire)) {
            if (c->seg->poisoned.load() || mono_ns() - t0 > c->deadline_ns) {
                c->fail("peer " + std::to_string(r) + " never published its window");
                return bail(ACL_ERR_INTERNAL);
            }
            usleep(1000);
        }
        if (c->seg->window_bytes[r] != c->window_bytes) {
            c->fail("peer " + std::to_string(r) + " opened with another window size");
            return bail(ACL_ERR_INTERNAL);
        }
        void *p = nullptr;
        if (c->hip(hipIpcOpenMemHandle(&p, c->seg->window[r], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) return bail(ACL_ERR_INTERNAL);
        c->win[r] = (uint8_t *)p;
    }
    if (c->barrier()) return bail(ACL_ERR_INTERNAL);  // everybody has mapped everybody
    if (rank == 0) shm_unlink(shm_name);               // (the mappings keep the segment alive; the name is free again)
    *out = c;
    return ACL_OK;
}

void aclipc_comm(aclipc *c, acl_shard_comm_t *out) {
    out->user = c;
    out->all_gather = cb_all_gather;
    out->all_reduce_max_u8 = cb_all_reduce;
    out->all_to_all = cb_all_to_all;
}

void aclipc_stats(aclipc *c, uint64_t out[4]) { std::memcpy(out, c->stat, sizeof c->stat); }

// a plain rendezvous of the ranks (tests: "both processes have loaded their graph")
int aclipc_barrier(aclipc *c) { return c->barrier() ? ACL_ERR_INTERNAL : ACL_OK; }

void aclipc_close(aclipc *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (uint32_t r = 0; r < c->world; r++)
        if (r != c->rank && c->win[r]) (void)hipIpcCloseMemHandle(c->win[r]);
    // the peers may still have this window mapped: they close their mapping first (a barrier would be the polite way, but close must also
    // work when a peer is dead), and the runtime keeps the allocation alive until the last mapping is gone
    if (c->mine) (void)hipFree(c->mine);
    if (c->seg) munmap(c->seg, sizeof(Segment));
    delete c;
}

}  // extern "C"
