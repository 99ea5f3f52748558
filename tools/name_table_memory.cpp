// tools/name_table_memory.cpp -- host memory of the name tables at config-5 scale: N names of ONE type interned through the C ABI
// on a store-only engine (no GPU needed); prints resident memory per name and the rate.  build: g++ -O2 -std=c++17 tools/name_table_memory.cpp -I include
// -L spicedb-kubeapi-proxy_amd/lib -laclgpu -Wl,-rpath,... -o tools/bin/name_table_memory     usage: name_table_memory [names = 10000000]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "aclgpu.h"

static double rss_mib() {
    FILE *f = fopen("/proc/self/statm", "r");
    long pages = 0, res = 0;
    if (f && fscanf(f, "%ld %ld", &pages, &res) != 2) res = 0;
    if (f) fclose(f);
    return res * 4096.0 / (1 << 20);
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 10000000;
    acl_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.flags = ACL_FLAG_STORE_ONLY;
    acl_engine_t *h = nullptr;
    if (acl_open(&cfg, &h)) { fprintf(stderr, "acl_open: %s\n", acl_last_error()); return 1; }
    const char *schema = "definition user {}\ndefinition pod {\n relation viewer: user\n permission view = viewer\n}\n";
    if (acl_load_bootstrap(h, schema, strlen(schema), nullptr, 0)) { fprintf(stderr, "schema: %s\n", acl_last_error()); return 1; }
    const int tp = acl_type_id(h, "pod");
    const double r0 = rss_mib();
    const auto t0 = std::chrono::steady_clock::now();
    char nm[64];
    uint32_t id = 0;
    for (long i = 0; i < n; i++) {
        snprintf(nm, sizeof nm, "namespace-%ld/pod-%ld", i % 5000, i);  // ~25 bytes: inside a slot's inline bytes, beyond std::string's small buffer
        if (acl_intern(h, tp, nm, &id)) { fprintf(stderr, "intern: %s\n", acl_last_error()); return 1; }
        if (i + 1 == n / 2 || i + 1 == n)
            printf("%ld names: RSS +%.0f MiB = %.0f B per name (%.1f s)\n", i + 1, rss_mib() - r0, (rss_mib() - r0) * (1 << 20) / (i + 1),
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    // every name is found again
    long bad = 0;
    for (long i = 0; i < n; i += 997) {
        snprintf(nm, sizeof nm, "namespace-%ld/pod-%ld", i % 5000, i);
        uint32_t got = 0;
        if (acl_find(h, tp, nm, &got) || got != (uint32_t)i) bad++;
    }
    printf("lookups of every 997th name: %ld wrong\n", bad);
    acl_close(h);
    return bad != 0;
}
