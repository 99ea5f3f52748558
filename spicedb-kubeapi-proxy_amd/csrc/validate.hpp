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
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

namespace acl {

inline bool valid_object_id_bytewise(std::string_view s) {  // (the definition; what the vector form below is checked against, tests/test_store_cpu.py)
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

#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
// 16 bytes per step: a PostFilter call validates one resource id per pair, and a byte-at-a-time table walk was half of what its host pass cost per pair
// (13 of 25 ns on a 15-byte `namespace/name`).  Every load lies inside the string: ids of 16 bytes and more end with an overlapping block, shorter ones are
// assembled from two overlapping 8- or 4-byte loads.  Bytes >= 0x80 are negative in the signed compares and fall out of every range.
inline bool object_id_bytes_ok(__m128i c) {
    const __m128i lower = _mm_or_si128(c, _mm_set1_epi8(0x20));  // A-Z -> a-z; nothing else lands in a-z
    const __m128i alpha = _mm_and_si128(_mm_cmpgt_epi8(lower, _mm_set1_epi8('a' - 1)), _mm_cmplt_epi8(lower, _mm_set1_epi8('z' + 1)));
    const __m128i digit = _mm_and_si128(_mm_cmpgt_epi8(c, _mm_set1_epi8('/' - 1)), _mm_cmplt_epi8(c, _mm_set1_epi8('9' + 1)));  // '/' is '0' - 1
    const __m128i punct = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(c, _mm_set1_epi8('_')), _mm_cmpeq_epi8(c, _mm_set1_epi8('|'))),
                                       _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(c, _mm_set1_epi8('-')), _mm_cmpeq_epi8(c, _mm_set1_epi8('='))), _mm_cmpeq_epi8(c, _mm_set1_epi8('+'))));
    return _mm_movemask_epi8(_mm_or_si128(_mm_or_si128(alpha, digit), punct)) == 0xFFFF;
}
inline bool valid_object_id(std::string_view s) {
    const size_t n = s.size();
    const char *p = s.data();
    if (n == 0 || n > 1024) return false;
    if (n >= 16) {
        for (size_t i = 0; i + 16 <= n; i += 16)
            if (!object_id_bytes_ok(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i)))) return false;
        return (n & 15u) == 0 || object_id_bytes_ok(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + n - 16)));
    }
    if (n >= 8) {
        long long a, b;
        __builtin_memcpy(&a, p, 8);
        __builtin_memcpy(&b, p + n - 8, 8);
        return object_id_bytes_ok(_mm_set_epi64x(b, a));
    }
    if (n >= 4) {
        int a, b;
        __builtin_memcpy(&a, p, 4);
        __builtin_memcpy(&b, p + n - 4, 4);
        return object_id_bytes_ok(_mm_set_epi32(b, a, b, a));
    }
    return valid_object_id_bytewise(s);
}
#else
inline bool valid_object_id(std::string_view s) { return valid_object_id_bytewise(s); }
#endif

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
