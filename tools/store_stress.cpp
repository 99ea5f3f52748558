// store_stress -- the host side of the engine under concurrent callers, for ThreadSanitizer (no GPU: a store-only engine).
// Writers (WriteRelationships with TOUCH / DELETE / CREATE + preconditions, activity.go:47-149), readers (ReadRelationships, name lookups,
// the Watch feed), the snapshot self-checks (in-place patch, background compaction) and single checks through the micro-batcher (refused:
// no GPU) all at once.  Exit code 0 = every call returned what it must; run under TSan for the races:
//   hipcc/clang++ -fsanitize=thread over csrc/*.cpp and this file (tools/tsan.sh)
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
    "definition group {\n  relation member: user | group#member\n}\n"
    "definition namespace {\n  relation viewer: user | group#member\n  relation creator: user\n  permission view = viewer + creator\n}\n"
    "definition pod {\n  relation namespace: namespace\n  relation viewer: user | group#member\n  relation creator: user\n"
    "  permission view = viewer + creator + namespace->view\n}\n"
    // the dual write's short-lived objects (workflow.go:392-462): their ids are recycled under everybody else (quarantine 0 below)
    "definition lock {\n  relation workflow: workflow\n}\n"
    "definition workflow {}\n";

// (every failed expectation says where: the harness runs many shapes at once and a count alone does not say which one)
#define BAD() (fprintf(stderr, "expectation failed: store_stress.cpp:%d\n", __LINE__), bad.fetch_add(1))

int main(int argc, char **argv) {
    const int ROUNDS = argc > 1 ? atoi(argv[1]) : 300;
    // "reload": a thread re-runs acl_load_bootstrap (same schema and relationships) under everybody else; the expectations that a reload
    // between two calls of one thread can void are then not counted -- the run is for crashes and ThreadSanitizer reports
    const bool reload = argc > 2 && !strcmp(argv[2], "reload");
    // "restart": a thread stops and restarts the micro-batcher under the single-check callers (a request appended while
    // acl_batcher_stop ran could be left behind in its queue for ever -- a lost request shows here as a caller that never returns)
    const bool restart = argc > 2 && !strcmp(argv[2], "restart");
    acl_engine_t *h = nullptr;
    acl_config_t cfg{-1, 0, 0, ACL_FLAG_STORE_ONLY, 0, 0};
    if (acl_open(&cfg, &h)) return 1;
    std::string rels;
    for (int p = 0; p < 2000; p++) {
        char b[200];
        snprintf(b, sizeof b, "pod:ns%d/p%d#namespace@namespace:ns%d\npod:ns%d/p%d#viewer@user:u%d\n", p % 50, p, p % 50, p % 50, p, p % 300);
        rels += b;
    }
    for (int g = 0; g < 100; g++) {
        char b[160];
        snprintf(b, sizeof b, "group:g%d#member@user:u%d\ngroup:g%d#member@group:g%d#member\nnamespace:ns%d#viewer@group:g%d#member\n", g, g, g, (g + 1) % 100, g % 50, g);
        rels += b;
    }
    setenv("ACL_ID_QUARANTINE_MS", "0", 1);  // (read when a schema is loaded) freed ids are taken by the next new name at once
    if (acl_load_bootstrap(h, kSchema, strlen(kSchema), rels.data(), rels.size())) { fprintf(stderr, "load: %s\n", acl_last_error()); return 1; }
    if (acl_batcher_start(h, 256, 50)) return 1;
    std::atomic<int> bad{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    // writers
    for (int wtr = 0; wtr < 2; wtr++)
        th.emplace_back([&, wtr] {
            unsigned s = 77u + wtr;
            auto rnd = [&](unsigned m) { s = s * 1664525u + 1013904223u; return (s >> 8) % m; };
            for (int r = 0; r < ROUNDS; r++) {
                char rid[64], sid[32];
                const int p = (int)rnd(4000);  // half of them new pods
                snprintf(rid, sizeof rid, "ns%d/p%d", p % 50, p);
                snprintf(sid, sizeof sid, "u%d", (int)rnd(400));
                acl_update_t u[2] = {{(int32_t)(r % 3 == 2 ? ACL_OP_DELETE : ACL_OP_TOUCH), {"pod", rid, "viewer", "user", sid, "", 0}},
                                     {ACL_OP_TOUCH, {"pod", rid, "creator", "user", sid, "", 0}}};
                uint64_t rev = 0;
                if (acl_write(h, u, 2, nullptr, 0, &rev) || !rev) BAD();
                {   // W1 / W2 of a kube write: a lock behind MUST_NOT_MATCH, deleted again -- its id (and the workflow's) goes round
                    char lk[48], wf[48];
                    snprintf(lk, sizeof lk, "l%d-%d", wtr, r);
                    snprintf(wf, sizeof wf, "w%d-%d", wtr, r);
                    acl_filter_t pre{ACL_PRE_MUST_NOT_MATCH, "lock", lk, "workflow", "workflow", nullptr, nullptr};
                    acl_update_t c{ACL_OP_CREATE, {"lock", lk, "workflow", "workflow", wf, "", 0}};
                    if (acl_write(h, &c, 1, &pre, 1, &rev) && !reload) BAD();
                    acl_update_t d{ACL_OP_DELETE, {"lock", lk, "workflow", "workflow", wf, "", 0}};
                    if (acl_write(h, &d, 1, nullptr, 0, &rev) && !reload) BAD();
                }
                if (r % 16 == 0) {  // CREATE of something that exists must fail, atomically (activity.go:62-74)
                    acl_update_t c{ACL_OP_CREATE, {"pod", rid, "creator", "user", sid, "", 0}};
                    if (acl_write(h, &c, 1, nullptr, 0, &rev) != ACL_ERR_ALREADY_EXISTS && !reload) BAD();
                    acl_filter_t pre{ACL_PRE_MUST_MATCH, "pod", rid, "creator", "user", sid, nullptr};
                    acl_update_t t{ACL_OP_TOUCH, {"pod", rid, "viewer", "group", "g1", "member", 0}};
                    if (acl_write(h, &t, 1, &pre, 1, &rev) && !reload) BAD();
                }
            }
        });
    // readers: ReadRelationships, name lookups, the Watch feed
    for (int rd = 0; rd < 2; rd++)
        th.emplace_back([&, rd] {
            uint64_t cursor = 0;
            acl_watch_poll(h, UINT64_MAX, nullptr, 0, nullptr, nullptr, &cursor);
            const int tp = acl_type_id(h, "pod"), tu = acl_type_id(h, "user");
            unsigned s = 5u + rd;
            while (!stop.load()) {
                s = s * 1664525u + 1013904223u;
                char rid[64];
                const int p = (int)((s >> 8) % 2000);
                snprintf(rid, sizeof rid, "ns%d/p%d", p % 50, p);
                acl_filter_t f{0, "pod", rid, nullptr, nullptr, nullptr, nullptr};
                long n = 0;
                if (acl_read(h, &f, [](void *u, const acl_relationship_t *r) { if (r->resource_id) ++*(long *)u; }, &n) || n < 1) BAD();  // (its namespace row is never deleted)
                uint32_t id = 0;
                if (acl_find(h, tp, rid, &id)) BAD();
                else if (!reload) {  // (a reload renumbers: the id may name another object by the time the name is asked for)
                    if (const char *nm = acl_object_name(h, tp, id)) { if (strcmp(nm, rid) != 0) BAD(); }
                }
                acl_intern(h, tu, ("fresh" + std::to_string(s % 5000)).c_str(), &id);
                uint64_t next = 0;
                long seen = 0;
                const int rc = acl_watch_poll(h, cursor, &tp, 1, [](void *u, uint64_t, int32_t, const acl_relationship_t *) { ++*(long *)u; }, &seen, &next);
                if (rc == ACL_OK) cursor = next;
                else if (rc == ACL_ERR_OUT_OF_RANGE) acl_watch_poll(h, UINT64_MAX, nullptr, 0, nullptr, nullptr, &cursor);
                else BAD();
            }
        });
    // a Watch stream as the shim holds it: blocked in acl_watch_wait (a condition variable behind the write path), then a poll; a cursor that an
    // id's recycling has overtaken is refused (OUT_OF_RANGE) and the stream starts over at the head
    th.emplace_back([&] {
        uint64_t cursor = 0, head = 0;
        acl_watch_poll(h, UINT64_MAX, nullptr, 0, nullptr, nullptr, &cursor);
        const int tl = acl_type_id(h, "lock");
        long seen = 0, waits = 0;
        while (!stop.load()) {
            acl_call_opts_t o{nullptr, 5 * 1000 * 1000};  // 5 ms
            const int wrc = acl_watch_wait(h, cursor, &tl, 1, &o, &head);
            if (wrc && wrc != ACL_ERR_DEADLINE_EXCEEDED && wrc != ACL_ERR_OUT_OF_RANGE) BAD();
            waits++;
            uint64_t next = 0;
            const int rc = acl_watch_poll(h, cursor, &tl, 1, [](void *u, uint64_t, int32_t, const acl_relationship_t *) { ++*(long *)u; }, &seen, &next);
            if (rc == ACL_OK) cursor = next;
            else if (rc == ACL_ERR_OUT_OF_RANGE) acl_watch_poll(h, UINT64_MAX, nullptr, 0, nullptr, nullptr, &cursor);
            else BAD();
        }
        if (!waits && ROUNDS >= 100) BAD();  // (a short run may be over before this thread is scheduled at all)
    });
    // snapshot maintenance: the patcher and the background compaction's two halves, verified against the store each time
    th.emplace_back([&] {
        int k = 0;
        while (!stop.load()) {
            int patched = 0, adopted = 0;
            if (acl_selfcheck_snapshot(h, &patched)) { fprintf(stderr, "selfcheck: %s\n", acl_last_error()); BAD(); }
            if (++k % 4 == 0) {
                if (acl_selfcheck_compaction(h, 0, &adopted)) { fprintf(stderr, "compaction 0: %s\n", acl_last_error()); BAD(); }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                const int rc1 = acl_selfcheck_compaction(h, 1, &adopted);  // (a bootstrap reload in between drops the build: "phase 1 without phase 0")
                if (rc1 && !(reload && rc1 == ACL_ERR_FAILED_PRECONDITION)) { fprintf(stderr, "compaction 1: %s\n", acl_last_error()); BAD(); }
            }
        }
    });
    if (reload)
        th.emplace_back([&] {
            while (!stop.load()) {
                if (acl_load_bootstrap(h, kSchema, strlen(kSchema), rels.data(), rels.size())) BAD();
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
    if (restart)
        th.emplace_back([&] {
            while (!stop.load()) {
                if (acl_batcher_stop(h)) BAD();
                std::this_thread::sleep_for(std::chrono::microseconds(200));
                if (acl_batcher_start(h, 256, 50)) BAD();
                std::this_thread::sleep_for(std::chrono::microseconds(700));
            }
        });
    // bulk name resolution (acl_resolve_bulk_v: the string entry points' host half, works store-only) under the writers -- 6 000 items go through the
    // interning pool, 300 stay on the calling thread; a pod's id must be the one acl_find gives (its namespace row is never deleted, so the id is never
    // recycled), whatever the writers intern or recycle meanwhile
    th.emplace_back([&] {
        const int tp = acl_type_id(h, "pod");
        std::vector<std::string> names;
        for (int p = 0; p < 2000; p++) names.push_back("ns" + std::to_string(p % 50) + "/p" + std::to_string(p));
        std::vector<acl_check_item_v_t> items(6000);
        std::vector<acl_item_t> out(items.size());
        std::vector<int32_t> err(items.size());
        std::vector<std::string> subj(items.size());
        unsigned s = 4242u;
        while (!stop.load()) {
            for (size_t i = 0; i < items.size(); i++) {
                s = s * 1664525u + 1013904223u;
                const std::string &r = names[(s >> 8) % names.size()];
                subj[i] = (i % 7 == 3 ? "fresh" : "u") + std::to_string((s >> 12) % 400);
                items[i] = acl_check_item_v_t{{"pod", 3}, {r.data(), r.size()}, {"view", 4}, {"user", 4}, {subj[i].data(), subj[i].size()}, {nullptr, 0}};
            }
            for (size_t n : {items.size(), (size_t)300}) {
                if (acl_resolve_bulk_v(h, items.data(), n, out.data(), err.data())) { BAD(); continue; }
                for (size_t i = 0; i < n; i += 97) {
                    uint32_t id = 0;
                    if (err[i] || out[i].resource_type != tp) BAD();
                    else if (!reload && (acl_find(h, tp, std::string(items[i].resource_id.p, items[i].resource_id.n).c_str(), &id) || id != out[i].resource_id)) BAD();
                }
            }
        }
    });
    // single checks: strings interned under the shared name lock, queued, refused by the pass (no GPU)
    for (int c = 0; c < 4; c++)
        th.emplace_back([&, c] {
            unsigned s = 1000u + c;
            acl_completion_t comp[16];
            while (!stop.load()) {
                s = s * 1664525u + 1013904223u;
                char rid[64], sid[32];
                const int p = (int)((s >> 8) % 2000);
                snprintf(rid, sizeof rid, "ns%d/p%d", p % 50, p);
                snprintf(sid, sizeof sid, "u%d", (int)((s >> 12) % 300));
                acl_check_item_t it{"pod", rid, "view", "user", sid, ""};
                uint8_t perm = 9;
                int32_t err = 0;
                if (c & 1) {
                    if (acl_check_one(h, &it, &perm, &err) != ACL_ERR_UNAVAILABLE) BAD();
                } else {
                    const int src = acl_check_one_submit(h, &it, s);
                    if (src && !(restart && src == ACL_ERR_FAILED_PRECONDITION)) BAD();  // (no batcher at this instant: the shim falls back to a blocking call)
                    size_t k = 0;
                    if (acl_check_completions(h, comp, 16, 1000000, &k)) BAD();
                    for (size_t j = 0; j < k; j++)
                        if (comp[j].rc != ACL_ERR_UNAVAILABLE || comp[j].perm != ACL_PERM_UNSPECIFIED) BAD();
                }
                uint8_t pb[4];
                int32_t eb[4];
                acl_check_item_t four[4] = {it, it, it, it};
                if (acl_check_bulk(h, four, 4, pb, eb) != ACL_ERR_UNAVAILABLE) BAD();
            }
        });
    for (int i = 0; i < 2; i++) th[i].join();
    stop = true;
    for (size_t i = 2; i < th.size(); i++) th[i].join();
    acl_batcher_stop(h);
    int patched = 0;
    if (acl_selfcheck_snapshot(h, &patched)) BAD();
    acl_close(h);
    printf("store_stress%s: %d rounds per writer, %d failed expectations\n", restart ? " (batcher restarts)" : "", ROUNDS, bad.load());
    return bad.load() ? 1 : 0;
}
