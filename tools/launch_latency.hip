// tools/launch_latency.hip -- what a small call's floor is made of on this box: hipLaunchKernelGGL's own time, the time until a
// flag the kernel stores into pinned host memory becomes visible to a spinning host thread, and the time until hipStreamSynchronize returns.
// build: hipcc --offload-arch=gfx950 -O2 tools/launch_latency.hip -o tools/bin/launch_latency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_flag(volatile unsigned *flag, unsigned v, const unsigned *src, unsigned *sink, int hops) {
    unsigned x = threadIdx.x;
    for (int i = 0; i < hops; i++) x = src[x & 1023u];  // `hops` dependent gathers: a stand-in for a walk's chain of trips
    if (threadIdx.x == 0) {
        sink[0] = x;
        __threadfence_system();
        *flag = v;
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *flag, *dflag, *src, *sink;
    hipHostMalloc((void **)&flag, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void **)&dflag, flag, 0);
    hipMalloc((void **)&src, 4096);
    hipMalloc((void **)&sink, 64);
    hipMemset(src, 0, 4096);
    for (int hops : {0, 8, 24}) {
        std::vector<double> tl, tf, ts;
        for (unsigned it = 1; it <= 2200; it++) {
            *flag = 0;
            const double t0 = now_us();
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dflag, it, src, sink, hops);
            const double t1 = now_us();
            while (*(volatile unsigned *)flag != it) {}
            const double t2 = now_us();
            hipStreamSynchronize(s);
            const double t3 = now_us();
            if (it > 200) { tl.push_back(t1 - t0); tf.push_back(t2 - t0); ts.push_back(t3 - t0); }
        }
        // and the plain form: launch + synchronise, no spinning
        std::vector<double> tp;
        for (unsigned it = 1; it <= 2200; it++) {
            const double t0 = now_us();
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dflag, it, src, sink, hops);
            hipStreamSynchronize(s);
            if (it > 200) tp.push_back(now_us() - t0);
        }
        auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("hops %2d: launch call %.1f us | flag visible to a spinning host %.1f us | + synchronize returned %.1f us | launch + synchronize alone %.1f us (p50 of 2000)\n", hops, med(tl), med(tf),
               med(ts), med(tp));
    }
    return 0;
}
