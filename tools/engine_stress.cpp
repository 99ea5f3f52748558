// engine_stress -- every concurrent call shape of the seam at once against ONE engine on a GPU, answers compared with the same calls made
// alone: chip-filling batches from three blocking callers (the chained path), a submit/wait window, 64-item batches, single checks through
// the micro-batcher (blocking and completion queue), string batches, PostFilter calls for one user, LookupResources, and a writer whose relationships touch only pods no request names
// (so every answer must stay what it was) but force snapshot patches and background compactions under the readers.
//   g++ -O2 -std=c++17 tools/engine_stress.cpp -Iinclude -Lspicedb-kubeapi-proxy_amd/lib -laclgpu -lpthread -o tools/bin/engine_stress
// (tools/tsan.sh also builds it instrumented, but ThreadSanitizer on a GPU box only reports the uninstrumented HIP / HSA runtimes' own
// accesses -- profiles/r02_engine_stress.txt; the host side is covered on store-only engines.)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "aclgpu.h"

static const char *kSchema =
    "definition user {}\n"
    "definition namespace {\n  relation viewer: user\n  relation creator: user\n  permission view = viewer + creator\n}\n"
    "definition pod {\n  relation namespace: namespace\n  relation viewer: user\n  relation creator: user\n"
    "  permission view = viewer + creator + namespace->view\n}\n"
    // (a RECURSIVE corner of the schema, which neither the writer nor the other requests touch: one user's doc pairs through acl_check_bulk_v take the reverse
    //  walk only after a forward sweep has shown that no doc's Check ends at the depth limit on the snapshot at hand -- engine.cpp no_object_is_deep)
    "definition group {\n  relation member: user | group#member\n}\n"
    "definition doc {\n  relation viewer: user | group#member\n  permission view = viewer\n}\n";

int main(int argc, char **argv) {
    const double SECONDS = argc > 1 ? atof(argv[1]) : 3.0;
    const int NPOD = 100000, NREQ_POD = 90000 /* requests name pods below this; the writer works above it */, NNS = 1000, NUSER = 10000;
    acl_engine_t *h = nullptr;
    acl_config_t cfg{-1, 0, 0, 0, 6, 0};
    if (acl_open(&cfg, &h)) { fprintf(stderr, "acl_open: %s\n", acl_last_error()); return 2; }
    std::string rels;
    unsigned s = 4242;
    auto rnd = [&](unsigned m) { s = s * 1664525u + 1013904223u; return (s >> 8) % m; };
    for (int p = 0; p < NPOD; p++) {
        char b[200];
        const int ns = p % NNS;
        snprintf(b, sizeof b, "pod:ns%d/p%d#namespace@namespace:ns%d\npod:ns%d/p%d#creator@user:u%d\npod:ns%d/p%d#viewer@user:u%d\n", ns, p, ns, ns, p, (int)rnd(NUSER), ns, p, (int)rnd(NUSER));
        rels += b;
    }
    for (int n = 0; n < NNS; n++)
        for (int k = 0; k < 10; k++) { char b[96]; snprintf(b, sizeof b, "namespace:ns%d#viewer@user:u%d\n", n, (int)((n * 31 + k * 977) % NUSER)); rels += b; }
    const int NDOC = 20000, NGROUP = 200;
    for (int g = 0; g < NGROUP; g++) {
        char b[160];
        if (g + 50 < NGROUP) { snprintf(b, sizeof b, "group:g%d#member@group:g%d#member\ngroup:g%d#member@group:g%d#member\n", g, g + 50, g, g + 1 + (g * 7 + 3) % 49); rels += b; }  // (children above their parent: no cycle)
        for (int k = 0; k < 6; k++) { snprintf(b, sizeof b, "group:g%d#member@user:u%d\n", g, (g * 13 + k * 101) % 64); rels += b; }
    }
    for (int d = 0; d < NDOC; d++) {
        char b[160];
        snprintf(b, sizeof b, "doc:d%d#viewer@group:g%d#member\ndoc:d%d#viewer@user:u%d\n", d, (int)rnd(NGROUP), d, (int)rnd(64));
        rels += b;
    }
    if (acl_load_bootstrap(h, kSchema, strlen(kSchema), rels.data(), rels.size())) { fprintf(stderr, "load: %s\n", acl_last_error()); return 2; }
    if (acl_snapshot(h)) { fprintf(stderr, "snapshot: %s\n", acl_last_error()); return 2; }
    const int tp = acl_type_id(h, "pod"), tu = acl_type_id(h, "user"), pv = acl_relation_id(h, tp, "view");
    std::vector<uint32_t> pod_id(NREQ_POD), user_id(NUSER);
    for (int p = 0; p < NREQ_POD; p++) { char nm[64]; snprintf(nm, sizeof nm, "ns%d/p%d", p % NNS, p); if (acl_find(h, tp, nm, &pod_id[p])) return 2; }
    for (int u = 0; u < NUSER; u++) { char nm[32]; snprintf(nm, sizeof nm, "u%d", u); if (acl_intern(h, tu, nm, &user_id[u])) return 2; }
    auto make = [&](size_t n, unsigned seed) {
        std::vector<acl_item_t> v(n);
        unsigned q = seed;
        for (auto &it : v) {
            q = q * 1664525u + 1013904223u;
            const uint32_t p = (q >> 8) % NREQ_POD;
            q = q * 1664525u + 1013904223u;
            it = acl_item_t{(uint16_t)tp, (uint16_t)pv, pod_id[p], (uint16_t)tu, ACL_NO_RELATION, user_id[(q >> 8) % NUSER]};
        }
        return v;
    };
    struct Job { std::vector<acl_item_t> items; std::vector<uint8_t> want; };
    auto job = [&](size_t n, unsigned seed) {
        Job j{make(n, seed), std::vector<uint8_t>(n)};
        std::vector<int32_t> err(n);
        if (acl_check_bulk_ids(h, j.items.data(), n, j.want.data(), err.data())) { fprintf(stderr, "reference pass: %s\n", acl_last_error()); exit(2); }
        return j;
    };
    std::vector<Job> big, mid, small;
    for (int i = 0; i < 3; i++) big.push_back(job(200000, 11 + i));
    for (int i = 0; i < 2; i++) mid.push_back(job(65536, 21 + i));
    for (int i = 0; i < 2; i++) small.push_back(job(64, 31 + i));
    // LookupResources of one user, alone
    const size_t words = (acl_object_count(h, tp) + 31) / 32 + 64;
    std::vector<uint32_t> bm0(words);
    uint64_t cnt0 = 0;
    if (acl_lookup_resources_ids(h, tp, pv, tu, -1, user_id[7], bm0.data(), words, &cnt0)) { fprintf(stderr, "lookup: %s\n", acl_last_error()); return 2; }
    if (acl_batcher_start(h, 1024, 50)) return 2;

    std::atomic<long> bad{0}, calls{0}, by[12] = {};  // by: big, submit window, 64-item, singles, string sets 0..3, keep sets 0..1, lookup, writer
#define BAD(leg) do { if (!bad++) fprintf(stderr, "first failure: leg %d: %s\n", (int)(leg), acl_last_error()); by[leg]++; } while (0)
    std::atomic<bool> stop{false};
    auto check = [&](const Job &j, const std::vector<uint8_t> &got) {
        const bool same = memcmp(j.want.data(), got.data(), got.size()) == 0;
        calls++;
        return same;
    };
    std::vector<std::thread> th;
    for (int i = 0; i < 3; i++)  // chip-filling batches, blocking (>= 131 072 items: chained on the device)
        th.emplace_back([&, i] {
            std::vector<uint8_t> p(big[i].items.size());
            std::vector<int32_t> e(p.size());
            while (!stop.load()) {
                if (acl_check_bulk_ids(h, big[i].items.data(), p.size(), p.data(), e.data())) { BAD(0); break; }
                if (!check(big[i], p)) BAD(0);
            }
        });
    th.emplace_back([&] {  // a submit / wait window of two
        std::vector<uint8_t> p[2] = {std::vector<uint8_t>(65536), std::vector<uint8_t>(65536)};
        std::vector<int32_t> e[2] = {std::vector<int32_t>(65536), std::vector<int32_t>(65536)};
        while (!stop.load()) {
            acl_ticket_t *t[2] = {nullptr, nullptr};
            for (int k = 0; k < 2; k++)
                if (acl_check_bulk_ids_submit(h, mid[k].items.data(), 65536, p[k].data(), e[k].data(), &t[k])) BAD(1);
            for (int k = 0; k < 2; k++) {
                if (t[k] && acl_ticket_wait(h, t[k])) BAD(1);
                else if (!check(mid[k], p[k])) BAD(1);
            }
        }
    });
    for (int i = 0; i < 2; i++)  // the proxy's own batch size
        th.emplace_back([&, i] {
            std::vector<uint8_t> p(64);
            std::vector<int32_t> e(64);
            while (!stop.load()) {
                if (acl_check_bulk_ids(h, small[i].items.data(), 64, p.data(), e.data())) { BAD(2); break; }
                if (!check(small[i], p)) BAD(2);
            }
        });
    for (int i = 0; i < 4; i++)  // single checks by name: blocking and through the completion queue
        th.emplace_back([&, i] {
            unsigned q = 900 + i;
            acl_completion_t comp[32];
            while (!stop.load()) {
                q = q * 1664525u + 1013904223u;
                const int k = (q >> 8) % 64;
                const acl_item_t &it0 = small[0].items[k];
                const char *pn = acl_object_name(h, tp, it0.resource_id), *un = acl_object_name(h, tu, it0.subject_id);
                if (!pn || !un) { BAD(3); break; }
                std::string pns(pn), uns(un);  // (engine-owned strings are only good until the next mutating call)
                acl_check_item_t it{"pod", pns.c_str(), "view", "user", uns.c_str(), ""};
                if (i & 1) {
                    uint8_t perm = 0;
                    int32_t err = 0;
                    if (acl_check_one(h, &it, &perm, &err) || err || perm != small[0].want[k]) BAD(3);
                    calls++;
                } else {
                    if (acl_check_one_submit(h, &it, (uint64_t)k)) BAD(3);
                    size_t n = 0;
                    if (acl_check_completions(h, comp, 32, 2000000, &n)) BAD(3);
                    for (size_t j = 0; j < n; j++)
                        if (comp[j].rc || comp[j].err || comp[j].perm != small[0].want[comp[j].tag]) BAD(3);
                    calls += (long)n;
                }
            }
        });
    // string batches (acl_check_bulk_v: the interning pool under concurrent callers, its workers polling or asleep) and PostFilter calls for one user
    // (acl_check_bulk_keep_v: one reverse walk + the pool's two passes, the names locked shared while the device walks) -- under the writer, who interns
    // new users and patches the snapshot meanwhile.  Expected answers: the id path's, taken alone before the threads start.
    struct Named { std::vector<std::string> rid, sid; std::vector<acl_check_item_v_t> v; std::vector<uint8_t> want; std::vector<uint32_t> off; };
    auto named = [&](size_t n, unsigned seed, int one_user) {
        Named N;
        Job j = job(n, seed);
        if (one_user >= 0) {
            for (auto &it : j.items) it.subject_id = user_id[one_user];
            std::vector<int32_t> err(n);
            if (acl_check_bulk_ids(h, j.items.data(), n, j.want.data(), err.data())) exit(2);
        }
        N.want = j.want;
        for (const auto &it : j.items) {
            N.rid.emplace_back(acl_object_name(h, tp, it.resource_id));
            N.sid.emplace_back(acl_object_name(h, tu, it.subject_id));
        }
        static const char kPod[] = "pod", kView[] = "view", kUser[] = "user";
        for (size_t i = 0; i < n; i++)
            N.v.push_back(acl_check_item_v_t{{kPod, 3}, {N.rid[i].data(), N.rid[i].size()}, {kView, 4}, {kUser, 4}, {one_user >= 0 ? N.sid[0].data() : N.sid[i].data(), N.sid[i].size()}, {nullptr, 0}});
        for (size_t i = 0; i <= n; i++) N.off.push_back((uint32_t)i);
        return N;
    };
    // one user's doc pairs (the recursive corner): expected answers by id, alone
    const int td = acl_type_id(h, "doc"), dv = acl_relation_id(h, td, "view");
    auto named_docs = [&](size_t n, unsigned seed, int user) {
        Named N;
        std::vector<acl_item_t> items(n);
        unsigned q = seed;
        for (auto &it : items) {
            q = q * 1664525u + 1013904223u;
            char nm[32];
            snprintf(nm, sizeof nm, "d%d", (int)((q >> 8) % NDOC));
            uint32_t id = 0;
            if (acl_find(h, td, nm, &id)) exit(2);
            N.rid.emplace_back(nm);
            it = acl_item_t{(uint16_t)td, (uint16_t)dv, id, (uint16_t)tu, ACL_NO_RELATION, user_id[user]};
        }
        N.want.resize(n);
        std::vector<int32_t> err(n);
        if (acl_check_bulk_ids(h, items.data(), n, N.want.data(), err.data())) exit(2);
        N.sid.emplace_back("u" + std::to_string(user));
        static const char kDoc[] = "doc", kView[] = "view", kUser[] = "user";
        for (size_t i = 0; i < n; i++) N.v.push_back(acl_check_item_v_t{{kDoc, 3}, {N.rid[i].data(), N.rid[i].size()}, {kView, 4}, {kUser, 4}, {N.sid[0].data(), N.sid[0].size()}, {nullptr, 0}});
        return N;
    };
    std::vector<Named> strs, keeps;
    strs.push_back(named_docs(2500, 44, 5));  // (recursive permission: forward until a sweep has shown the snapshot shallow, then by the reverse walk)
    strs.push_back(named(8192, 41, -1));
    strs.push_back(named(700, 42, -1));
    strs.push_back(named(3000, 43, 9));  // (one user's pairs: on this schema -- no recursion -- the string call itself takes the reverse walk)
    keeps.push_back(named(4096, 51, 7));
    keeps.push_back(named(1500, 52, 100));
    for (size_t i = 0; i < strs.size(); i++)
        th.emplace_back([&, i] {
            const Named &N = strs[i];
            std::vector<uint8_t> p(N.v.size());
            std::vector<int32_t> e(N.v.size());
            while (!stop.load()) {
                if (acl_check_bulk_v(h, N.v.data(), N.v.size(), p.data(), e.data())) { BAD(4 + (int)i); break; }
                if (memcmp(p.data(), N.want.data(), p.size()) != 0) BAD(4 + (int)i);
                for (int32_t x : e) if (x) { BAD(4 + (int)i); break; }
                calls++;
                if (i > 1) std::this_thread::sleep_for(std::chrono::microseconds(400));  // (this one finds the pool asleep)
            }
        });
    for (size_t i = 0; i < keeps.size(); i++)
        th.emplace_back([&, i] {
            const Named &N = keeps[i];
            std::vector<uint8_t> keep(N.v.size());
            while (!stop.load()) {
                if (acl_check_bulk_keep_v(h, N.v.data(), N.v.size(), N.off.data(), N.v.size(), keep.data())) { BAD(8 + (int)i); break; }
                for (size_t k = 0; k < keep.size(); k++)
                    if ((keep[k] != 0) != (N.want[k] == ACL_PERM_HAS_PERMISSION)) { BAD(8 + (int)i); break; }
                calls++;
                if (i) std::this_thread::sleep_for(std::chrono::microseconds(250));
            }
        });
    th.emplace_back([&] {  // LookupResources
        std::vector<uint32_t> bm(words);
        while (!stop.load()) {
            uint64_t cnt = 0;
            if (acl_lookup_resources_ids(h, tp, pv, tu, -1, user_id[7], bm.data(), words, &cnt)) { BAD(10); break; }
            // the writer only adds viewers u7 never is, on pods above NREQ_POD ... of other users: this user's set must not change
            if (cnt != cnt0 || memcmp(bm.data(), bm0.data(), (NREQ_POD / 32) * 4) != 0) BAD(10);
            calls++;
        }
    });
    th.emplace_back([&] {  // the writer: pods nobody asks about, users other than u7
        unsigned q = 5;
        long w = 0;
        while (!stop.load()) {
            q = q * 1664525u + 1013904223u;
            const int p = NREQ_POD + (int)((q >> 8) % (NPOD - NREQ_POD));
            char rid[64], sid[32];
            snprintf(rid, sizeof rid, "ns%d/p%d", p % NNS, p);
            snprintf(sid, sizeof sid, "u%d", 8 + (int)((q >> 12) % (NUSER - 8)));
            acl_update_t u{(int32_t)(w % 3 == 2 ? ACL_OP_DELETE : ACL_OP_TOUCH), {"pod", rid, "viewer", "user", sid, "", 0}};
            uint64_t rev = 0;
            if (acl_write(h, &u, 1, nullptr, 0, &rev)) BAD(11);
            w++;
            std::this_thread::sleep_for(std::chrono::microseconds(300));  // (plain grants: they do not touch what the depth sweep showed, Store::path_adds)
        }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds((long)(SECONDS * 1000)));
    stop = true;
    for (auto &t : th) t.join();
    acl_batcher_stop(h);
    acl_stats_t st;
    acl_stats(h, &st);
    printf("engine_stress: %.1f s, %ld calls checked, %ld wrong or failed; snapshot patches %llu, compactions %llu, single-launch passes %llu, PostFilter calls by reverse walk %llu, depth sweeps %llu\n", SECONDS, calls.load(), bad.load(),
           (unsigned long long)st.snapshot_patches, (unsigned long long)st.snapshot_compactions, (unsigned long long)st.local_passes, (unsigned long long)st.keep_route_calls,
           (unsigned long long)st.depth_sweeps);
    if (bad.load()) {
        printf("wrong or failed by leg (big, submit window, 64-item, singles, strings x 4, keeps x 2, lookup, writer):");
        for (auto &b : by) printf(" %ld", b.load());
        printf("\n");
    }
    acl_close(h);
    return bad.load() || !st.keep_route_calls ? 1 : 0;
}
