// list_stress -- the list-level filters' parallel host code under the sanitizers (no GPU: a store-only engine; acl_prefilter_response needs the name tables only).
// A PodList and a Table of a few MB with strings full of quotes, backslash runs, brackets and commas; acl_selfcheck_json_array over them with chunks from 64 bytes
// up (every answer must be the same spans), acl_prefilter_response (lists and tables) and acl_bitmap_test_names from several threads at once, each output compared
// with the first one's.  tools/tsan.sh-style builds:
//   hipcc ... -fsanitize=thread            (races between the pool's workers: items, spans, keep bytes, the spliced output)
//   hipcc ... -fsanitize=address,undefined (the 16- / 64-byte loads at chunk ends and body ends, the tail buffer, span arithmetic)
// usage: list_stress [items = 4000] [threads = 3] [rounds = 6]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "aclgpu.h"

static const char *kSchema = "definition user {}\ndefinition pod {\n  relation viewer: user\n  permission view = viewer\n}\n";

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4000, T = argc > 2 ? atoi(argv[2]) : 3, ROUNDS = argc > 3 ? atoi(argv[3]) : 6;
    acl_engine_t *h = nullptr;
    acl_config_t cfg{-1, 0, 0, ACL_FLAG_STORE_ONLY, 0, 0};
    if (acl_open(&cfg, &h)) { fprintf(stderr, "acl_open: %s\n", acl_last_error()); return 2; }
    if (acl_load_bootstrap(h, kSchema, strlen(kSchema), "", 0)) { fprintf(stderr, "load: %s\n", acl_last_error()); return 2; }
    const int tp = acl_type_id(h, "pod");
    std::vector<std::string> names;
    std::vector<uint32_t> ids(N);
    for (int i = 0; i < N; i++) {
        names.push_back("ns" + std::to_string(i % 13) + "/pod-" + std::to_string(i));
        if (acl_intern(h, tp, names.back().c_str(), &ids[i])) return 2;
    }
    const size_t words = (acl_object_count(h, tp) + 31) / 32 + 1;
    std::vector<uint32_t> bm(words, 0);
    for (int i = 0; i < N; i += 3) bm[ids[i] >> 5] |= 1u << (ids[i] & 31u);
    // ---- the bodies
    auto item = [&](int i, bool row) {
        std::string md = "{\"name\":\"pod-" + std::to_string(i) + "\",\"namespace\":\"ns" + std::to_string(i % 13) + "\",\"annotations\":{\"cfg\":\"{\\\"a\\\":[\\\"}\\\",\\\"]\\\",\\\"\\\\\\\\\\\"," +
                         std::to_string(i) + "]}\",\"pad\":\"" + std::string(200 + i % 190, 'z') + "\\\\\"},\"labels\":{\"k,\":\"[v]\"}}";
        if (row) return "{\"cells\":[\"pod-" + std::to_string(i) + "\",\"1/1\",\"},{\"],\"object\":{\"kind\":\"PartialObjectMetadata\",\"metadata\":" + md + "}}";
        return "{\"kind\":\"Pod\",\"metadata\":" + md + ",\"spec\":{\"containers\":[{\"name\":\"c\",\"args\":[\"{\",\"]\",\"\\\\\\\"\",\",\"],\"ports\":[80,-1,2.5e3]}]}}";
    };
    std::string list = "{\"kind\":\"PodList\",\"apiVersion\":\"v1\",\"metadata\":{\"resourceVersion\":\"1\"},\"items\":[", table = "{\"kind\":\"Table\",\"columnDefinitions\":[{\"name\":\"Name\"}],\"rows\":[";
    for (int i = 0; i < N; i++) {
        list += (i ? "," : "") + item(i, false);
        table += (i ? ",\n " : "") + item(i, true);
    }
    list += "]}";
    table += "],\"tail\":[1,{\"z\":\"]\"}]}";
    printf("list %.2f MB, table %.2f MB, %d items\n", list.size() / 1e6, table.size() / 1e6, N);
    // ---- the element index at every chunk size gives the same spans
    long bad = 0;
    for (const std::string *b : {&list, &table}) {
        const size_t arr_open = b->find(b == &list ? "\"items\":[" : "\"rows\":[") + (b == &list ? 8 : 7);
        std::vector<size_t> ref(2 * (size_t)N), got(2 * (size_t)N);
        size_t n0 = 0, c0 = 0;
        if (acl_selfcheck_json_array(h, b->data(), b->size(), arr_open, 0, ref.data(), (size_t)N, &n0, &c0) || n0 != (size_t)N) { fprintf(stderr, "index: %s (%zu elements)\n", acl_last_error(), n0); bad++; }
        for (size_t chunk : {(size_t)64, (size_t)128, (size_t)192, (size_t)4096, (size_t)65536, (size_t)1 << 20}) {
            size_t n = 0, c = 0;
            if (acl_selfcheck_json_array(h, b->data(), b->size(), arr_open, chunk, got.data(), (size_t)N, &n, &c) || n != n0 || c != c0 || got != ref) { fprintf(stderr, "chunk %zu: differs\n", chunk); bad++; }
        }
    }
    // ---- the consumers, from T threads at once
    struct Out { std::string body; uint64_t kept = 0, total = 0; };
    auto filter = [&](const std::string &b, int kind, Out *o) {
        char *ob = nullptr;
        size_t on = 0;
        if (acl_prefilter_response(h, tp, bm.data(), words, "{{namespacedName}}", kind, b.data(), b.size(), &ob, &on, &o->kept, &o->total)) return false;
        o->body.assign(ob, on);
        acl_free(ob);
        return true;
    };
    Out l0, t0;
    if (!filter(list, ACL_BODY_LIST, &l0) || !filter(table, ACL_BODY_TABLE, &t0)) { fprintf(stderr, "filter: %s\n", acl_last_error()); return 1; }
    if (l0.kept != (uint64_t)(N + 2) / 3 || l0.total != (uint64_t)N || t0.kept != l0.kept || t0.total != (uint64_t)N) { fprintf(stderr, "kept %llu / %llu of %llu\n", (unsigned long long)l0.kept, (unsigned long long)t0.kept, (unsigned long long)l0.total); bad++; }
    std::vector<const char *> cn;
    for (auto &s : names) cn.push_back(s.c_str());
    std::atomic<long> abad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            std::vector<uint8_t> allowed(N);
            for (int r = 0; r < ROUNDS; r++) {
                Out o;
                const bool tab = (t + r) & 1;
                if (!filter(tab ? table : list, tab ? ACL_BODY_TABLE : ACL_BODY_LIST, &o) || o.body != (tab ? t0 : l0).body || o.kept != l0.kept) abad++;
                if (acl_bitmap_test_names(h, tp, bm.data(), words, cn.data(), (size_t)N, allowed.data())) abad++;
                for (int i = 0; i < N; i++)
                    if ((allowed[i] != 0) != (i % 3 == 0)) { abad++; break; }
            }
        });
    for (auto &x : th) x.join();
    bad += abad.load();
    // a broken element in the middle fails the call
    std::string broken = list;
    broken[broken.find("\"kind\":\"Pod\"", broken.size() / 2) + 7] = 'x';
    Out ob;
    if (filter(broken, ACL_BODY_LIST, &ob)) { fprintf(stderr, "a broken body was accepted\n"); bad++; }
    printf("list_stress: %d threads x %d rounds, %ld wrong\n", T, ROUNDS, bad);
    acl_close(h);
    return bad ? 1 : 0;
}
