// validate.hpp -- request validation of the authzed API, restated.
//
// EXTERNAL, UNVERIFIED: these are the `validate` rules of the authzed.api.v1 messages (github.com/authzed/authzed-go v1.10.0, go.mod:6) that the
// embedded SpiceDB applies before it evaluates anything, written down from memory -- the module is not under /root/reference and cannot be
// fetched here.  They matter to the proxy because a request that fails them is answered InvalidArgument AS A WHOLE: the reference treats any
// error of CheckBulkPermissions as "deny everything asked" (pkg/authz/check.go:48-52), fails the list response on it (postfilter.go:134-137)
// and treats InvalidArgument of a write as unrecoverable (pkg/authz/distributedtx/workflow.go:115-119).  The one reference-held vector is the
// empty CheckPermissionRequest (pkg/proxy/options_test.go:101-102).  DESIGN.md 6 holds the table; tests/golden/kats.json the vectors (status
// `unverified`), tests/ref_cases.py replays them through the Go harness the day it can be built.
//
//   object type      ^([a-z][a-z0-9_]{1,61}[a-z0-9]/)*[a-z][a-z0-9_]{1,62}[a-z0-9]$   <= 128 bytes
//   relation / perm  ^[a-z][a-z0-9_]{1,62}[a-z0-9]$                                    <= 64 bytes   (subject relation: that, "" or "...")
//   object id        ^[a-zA-Z0-9/_|\-=+]{1,}$                                          <= 1024 bytes
//   `*`              only as the SUBJECT id of a relationship or of a relationship filter, never with a subject relation,
//                    never in a Check / LookupResources request, never as a resource id
// A type / relation name that the LOADED SCHEMA declares is accepted whatever its spelling: this engine's schema loader takes short names
// (`definition u {}`) that the real schema compiler refuses, and a request naming a declared thing must not fail on a rule the schema
// itself got past.  For every schema the real engine accepts the two readings coincide.
#pragma once
#include <string_view>

namespace acl {

inline bool valid_object_id(std::string_view s) {
    static const struct Table {
        bool ok[256] = {};
        Table() {
            for (int c = 'a'; c <= 'z'; c++) ok[c] = true;
            for (int c = 'A'; c <= 'Z'; c++) ok[c] = true;
            for (int c = '0'; c <= '9'; c++) ok[c] = true;
            for (unsigned char c : {'/', '_', '|', '-', '=', '+'}) ok[c] = true;
        }
    } t;
    if (s.empty() || s.size() > 1024) return false;
    for (unsigned char c : s)
        if (!t.ok[c]) return false;
    return true;
}

inline bool valid_name_segment(std::string_view s, size_t max_len) {  // [a-z][a-z0-9_]{1,max_len-2}[a-z0-9]
    if (s.size() < 3 || s.size() > max_len) return false;
    auto lower = [](char c) { return c >= 'a' && c <= 'z'; };
    auto digit = [](char c) { return c >= '0' && c <= '9'; };
    if (!lower(s.front()) || !(lower(s.back()) || digit(s.back()))) return false;
    for (char c : s.substr(1, s.size() - 2))
        if (!(lower(c) || digit(c) || c == '_')) return false;
    return true;
}
inline bool valid_relation_name(std::string_view s) { return valid_name_segment(s, 64); }
inline bool valid_type_name(std::string_view s) {
    if (s.empty() || s.size() > 128) return false;
    for (;;) {  // prefix segments (<= 63 chars each), then the name (<= 64)
        const size_t slash = s.find('/');
        if (slash == std::string_view::npos) return valid_name_segment(s, 64);
        if (!valid_name_segment(s.substr(0, slash), 63)) return false;
        s.remove_prefix(slash + 1);
    }
}

}  // namespace acl
