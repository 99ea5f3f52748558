#include "store.hpp"
#include "validate.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
using namespace acl;
struct View { const char *p; size_t n; };
int main() {
    const size_t N = 65536;
    std::string blob;
    std::vector<size_t> off(N + 1);
    for (size_t i = 0; i < N; i++) { off[i] = blob.size(); blob += "ns" + std::to_string((i * 7919) % 977) + "/pod-" + std::to_string((i * 104729) % 845000); }
    off[N] = blob.size();
    std::vector<View> v(N * 6);
    const char *c[5] = {"pod", "view", "user", "user-123", ""};
    for (size_t i = 0; i < N; i++) {
        v[i * 6 + 0] = {c[0], 3}; v[i * 6 + 1] = {blob.data() + off[i], off[i + 1] - off[i]}; v[i * 6 + 2] = {c[1], 4}; v[i * 6 + 3] = {c[2], 4}; v[i * 6 + 4] = {c[3], 8}; v[i * 6 + 5] = {nullptr, 0};
    }
    std::vector<uint64_t> hv(N);
    auto T = [] { return std::chrono::steady_clock::now(); };
    for (int rep = 0; rep < 3; rep++) {
        auto tm = T();
        size_t acc = 0;
        for (size_t i = 0; i < N; i++) for (int k = 0; k < 6; k++) acc += (size_t)v[i * 6 + k].p + v[i * 6 + k].n;
        auto tn = T();
        for (size_t i = 0; i < N; i++) acc += (unsigned char)v[i * 6 + 1].p[0] + (unsigned char)v[i * 6 + 1].p[v[i * 6 + 1].n - 1];
        auto t0 = T();
        printf("stream views %.1f ns, touch names %.1f ns (%zu) | ", std::chrono::duration<double, std::nano>(tn - tm).count() / N, std::chrono::duration<double, std::nano>(t0 - tn).count() / N, acc);
        size_t bad = 0;
        for (size_t i = 0; i < N; i++) bad += !valid_object_id(std::string_view(v[i * 6 + 1].p, v[i * 6 + 1].n));
        auto t1 = T();
        for (size_t i = 0; i < N; i++) hv[i] = ObjectTable::hash_of(std::string_view(v[i * 6 + 1].p, v[i * 6 + 1].n));
        auto t2 = T();
        size_t diff = 0;
        for (size_t i = 0; i < N; i++)
            for (int k : {0, 2, 3, 4, 5}) {
                const View &x = v[i * 6 + k], &y = v[k];
                diff += !(x.n == y.n && (x.p == y.p || (x.p && y.p && std::memcmp(x.p, y.p, x.n) == 0) || (x.n == 0 && (!x.p || !y.p))));
            }
        auto t3 = T();
        auto ns = [&](auto a, auto b) { return std::chrono::duration<double, std::nano>(b - a).count() / N; };
        printf("validate %.1f ns  hash %.1f ns  constants %.1f ns   (bad %zu diff %zu h %llx)\n", ns(t0, t1), ns(t1, t2), ns(t2, t3), bad, diff, (unsigned long long)hv[5]);
    }
}
