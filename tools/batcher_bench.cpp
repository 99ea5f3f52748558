// batcher_bench -- the proxy's call shape against the C ABI: T host threads each issuing single CheckPermission calls
// (pkg/authz/check.go:76-94: one goroutine per check expression; watch.go:50: one per update), with and without the
// micro-batching front-end (acl_batcher_start).  Plain C++ over include/aclgpu.h, no Python in the timed path.
//   g++ -O2 -std=c++17 tools/batcher_bench.cpp -Iinclude -Lspicedb-kubeapi-proxy_amd/lib -laclgpu -lpthread -o /tmp/batcher_bench
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "aclgpu.h"

static const char *kSchema =
    "definition user {}\n"
    "definition namespace {\n  relation viewer: user\n  relation creator: user\n  permission view = viewer + creator\n}\n"
    "definition pod {\n  relation namespace: namespace\n  relation viewer: user\n  relation creator: user\n"
    "  permission view = viewer + creator + namespace->view\n}\n";

int main(int argc, char **argv) {
    // usage: batcher_bench [PER_THREAD] [threads ...]     default: 2000 checks per thread at 64 256 1024 threads
    const int PER = argc > 1 ? atoi(argv[1]) : 2000, NPOD = 100000, NNS = 1000, NUSER = 10000;
    std::vector<int> sweep;
    for (int a = 2; a < argc; a++) sweep.push_back(atoi(argv[a]));
    if (sweep.empty()) sweep = {64, 256, 1024};
    acl_engine_t *h = nullptr;
    acl_config_t cfg{-1, 0, 0, 0, 4, 0};
    // ACL_BATCHER_SIM_PASS_US=<us>: no GPU -- a store-only engine, whose passes are REFUSED (UNAVAILABLE) after spinning that long: times the
    // queueing / wake-up machinery alone (engine_callers.cpp); every check counts as an error, as it should
    const bool sim = getenv("ACL_BATCHER_SIM_PASS_US") != nullptr;
    if (sim) cfg.flags |= ACL_FLAG_STORE_ONLY;
    if (acl_open(&cfg, &h)) { fprintf(stderr, "acl_open: %s\n", acl_last_error()); return 1; }
    std::string rels;
    unsigned s = 12345;
    auto rnd = [&](unsigned m) { s = s * 1664525u + 1013904223u; return (s >> 8) % m; };
    std::vector<int> pod_ns(NPOD);
    for (int p = 0; p < NPOD; p++) {
        char b[160];
        int ns = rnd(NNS);
        pod_ns[p] = ns;
        snprintf(b, sizeof b, "pod:ns%d/p%d#namespace@namespace:ns%d\npod:ns%d/p%d#creator@user:u%d\n", ns, p, ns, ns, p, rnd(NUSER));
        rels += b;
        const int v0 = rnd(NUSER);
        for (int k = 0; k < 3; k++) { snprintf(b, sizeof b, "pod:ns%d/p%d#viewer@user:u%d\n", ns, p, (v0 + k * 3331) % NUSER); rels += b; }  // distinct viewers
    }
    for (int n = 0; n < NNS; n++) {
        const int v0 = rnd(NUSER);
        for (int k = 0; k < 10; k++) { char b[96]; snprintf(b, sizeof b, "namespace:ns%d#viewer@user:u%d\n", n, (v0 + k * 977) % NUSER); rels += b; }
    }
    if (acl_load_bootstrap(h, kSchema, std::string(kSchema).size(), rels.data(), rels.size())) { fprintf(stderr, "load: %s\n", acl_last_error()); return 1; }
    acl_snapshot(h);
    // ---- small batches through the host-id call (acl_check_bulk_ids): the single-launch path's latency, no Python in the way
    if (!sim) {
        const int tp = acl_type_id(h, "pod"), tu = acl_type_id(h, "user"), pv = acl_relation_id(h, tp, "view");
        std::vector<acl_item_t> items(8192);
        for (auto &it : items) {
            const int p = rnd(NPOD);
            char nm[64];
            snprintf(nm, sizeof nm, "ns%d/p%d", pod_ns[p], p);
            uint32_t pid = 0, uid = 0;
            acl_find(h, tp, nm, &pid);
            snprintf(nm, sizeof nm, "u%d", (int)rnd(NUSER));
            acl_find(h, tu, nm, &uid);
            it = acl_item_t{(uint16_t)tp, (uint16_t)pv, pid, (uint16_t)tu, ACL_NO_RELATION, uid};
        }
        std::vector<uint8_t> perm(items.size());
        std::vector<int32_t> err(items.size());
        for (int sz : {1, 64, 256, 1024, 4096, 8192}) {
            std::vector<double> us;
            for (int k = 0; k < 220; k++) {
                auto t0 = std::chrono::steady_clock::now();
                if (acl_check_bulk_ids(h, items.data(), sz, perm.data(), err.data())) { fprintf(stderr, "check: %s\n", acl_last_error()); return 1; }
                if (k >= 20) us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            std::sort(us.begin(), us.end());
            printf("{\"small_batch_items\": %d, \"p50_us\": %.1f, \"p95_us\": %.1f, \"graph\": \"3-level pod graph, 500 k relationships\"}\n", sz, us[us.size() / 2], us[us.size() * 95 / 100]);
        }
    }
    for (int T : sweep) {
        // request strings prepared up front (the proxy has them from its rule templates)
        std::vector<std::vector<std::string>> pod(T), usr(T);
        for (int t = 0; t < T; t++)
            for (int i = 0; i < PER; i++) {
                int p = rnd(NPOD);
                pod[t].push_back("ns" + std::to_string(i % 4 ? pod_ns[p] : (int)rnd(NNS)) + "/p" + std::to_string(p));  // 3 in 4 known pods, the rest unknown ids
                usr[t].push_back("u" + std::to_string(rnd(NUSER)));
            }
        for (int mode = (T > 64 || sim ? 1 : 0); mode < 2; mode++) {  // "one pass per call" only at the smallest thread count (it is the slow baseline)
            if (mode == 1 && acl_batcher_start(h, 4096, 50)) { fprintf(stderr, "batcher: %s\n", acl_last_error()); return 1; }
            acl_stats_reset(h);
            uint64_t nb0 = 0;
            acl_batcher_stats(h, &nb0, nullptr);
            std::atomic<long> has{0}, bad{0};
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    for (int i = 0; i < PER; i++) {
                        acl_check_item_t it{"pod", pod[t][i].c_str(), "view", "user", usr[t][i].c_str(), ""};
                        uint8_t perm = 0;
                        int32_t err = 0;
                        if (acl_check_one(h, &it, &perm, &err) || err) bad++;
                        else if (perm == ACL_PERM_HAS_PERMISSION) has++;
                    }
                });
            for (auto &x : th) x.join();
            double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            acl_stats_t st;
            acl_stats(h, &st);
            uint64_t nb = 0, ni = 0;
            acl_batcher_stats(h, &nb, &ni);
            printf("{\"mode\": \"%s\", \"threads\": %d, \"checks\": %ld, \"checks_per_s\": %.0f, \"mean_latency_us\": %.1f, \"device_passes\": %llu, \"single_launch_passes\": %llu, \"batcher_passes\": %llu, \"has\": %ld, \"errors\": %ld, \"host_threads\": %u}\n",
                   mode ? "micro-batched (max 4096 items / 50 us)" : "one device pass per call", T, (long)T * PER, T * PER / el, el * 1e6 / PER,
                   (unsigned long long)st.check_passes, (unsigned long long)st.local_passes, (unsigned long long)(nb - nb0), has.load(), bad.load(), std::thread::hardware_concurrency());
            if (mode == 1) acl_batcher_stop(h);
        }
        // ---- the same T concurrent callers as T *logical* callers (goroutines in the proxy: parked in user space, not in the kernel), each
        // with ONE check outstanding, multiplexed over M OS threads the way a cgo shim would: acl_check_one_submit + acl_check_completions.
        // Whichever thread collects a caller's answer issues that caller's next check.
        {
            if (acl_batcher_start(h, 4096, 50)) { fprintf(stderr, "batcher: %s\n", acl_last_error()); return 1; }
            const int M = getenv("BENCH_POLLERS") ? atoi(getenv("BENCH_POLLERS")) : 8;
            uint64_t nb0 = 0, nb = 0;
            acl_batcher_stats(h, &nb0, nullptr);
            std::vector<int> progress(T, 0);  // (a caller has one check outstanding: only the thread holding its completion touches its slot)
            std::atomic<long> has{0}, bad{0}, finished{0};
            auto submit = [&](int t) {
                const int i = progress[t]++;
                acl_check_item_t it{"pod", pod[t][i].c_str(), "view", "user", usr[t][i].c_str(), ""};
                if (acl_check_one_submit(h, &it, (uint64_t)t)) bad++;
            };
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int m = 0; m < M; m++)
                th.emplace_back([&, m] {
                    for (int t = m; t < T; t += M) submit(t);
                    acl_completion_t c[64];
                    while (finished.load(std::memory_order_relaxed) < T) {
                        size_t k = 0;
                        if (acl_check_completions(h, c, 64, 1000000, &k)) { bad++; break; }
                        for (size_t j = 0; j < k; j++) {
                            if (c[j].rc || c[j].err) bad++;
                            else if (c[j].perm == ACL_PERM_HAS_PERMISSION) has++;
                            const int t = (int)c[j].tag;
                            if (progress[t] < PER) submit(t);
                            else finished++;
                        }
                    }
                });
            for (auto &x : th) x.join();
            double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            acl_batcher_stats(h, &nb, nullptr);
            printf("{\"mode\": \"completion queue (submit + poll, %d OS threads)\", \"threads\": %d, \"logical_callers\": %d, \"checks\": %ld, \"checks_per_s\": %.0f, \"mean_latency_us\": %.1f, \"batcher_passes\": %llu, \"has\": %ld, \"errors\": %ld}\n",
                   M, M, T, (long)T * PER, T * PER / el, el * 1e6 / PER, (unsigned long long)(nb - nb0), has.load(), bad.load());
            acl_batcher_stop(h);
        }
    }
    acl_close(h);
    return 0;
}
