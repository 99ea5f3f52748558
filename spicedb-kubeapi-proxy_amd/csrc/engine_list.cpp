// engine_list.cpp -- PostFilter at LIST level: the reference's filterListResponse (pkg/authz/postfilter.go:17-55) over the
// kube list response's bytes.
//
// Reference flow: json.Unmarshal(body) -> items[] -> for every item and every PostFilter template resolve
// `type:id#perm@type:id` from the item's metadata.name / metadata.namespace (postfilter.go:73-119) -> ONE
// CheckBulkPermissions (postfilter.go:134) -> keep the items whose pairs are all HAS_PERMISSION (postfilter.go:144-178)
// -> json.Marshal.  Here the body is scanned for the item spans and their metadata -- bodies from 512 KB on by all host threads: the spans through the parallel
// element index (json_index.hpp), the elements validated side by side (scan_array) --, the K x F resolved pairs go through acl_check_bulk_keep_v (one reverse
// walk per template + bit tests for the usual one-user list; else one forward device pass), and the answer is the ORIGINAL bytes with the dropped items' spans cut
// out -- no generic decode / re-encode of a body that can be many megabytes.  (The reference's re-marshal sorts object
// keys; the spliced document is the same JSON value for every key order, which is what kube clients parse.)
//
// Templates: the general rule language (Bloblang / CEL, pkg/rules) stays in Go (out of scope); this entry
// point renders the placeholder form every shipped rule uses (deploy/rules.yaml:68 `pod:{{namespacedName}}#view@user:{{user.name}}`):
// {{name}} {{namespace}} {{namespacedName}} {{user.name}}.  Rules that need more resolve in Go and call acl_check_bulk_keep.
//
// PreFilter consumers (acl_prefilter_response): the reference's filterList / filterTable / filterObject (pkg/authz/responsefilterer.go:
// 349-416) keep what prefilterResult.IsAllowed(namespace, name) admits (lookups.go:25-36) -- here the LookupResources bitmap tested by the
// object id the rule's inverse mapping gives an item, over the same scanned spans.
#include "engine_internal.hpp"
#include "json_index.hpp"

namespace {

struct Scanner {
    const char *p, *e;
    bool ok = true;
    void ws() {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool fail() {
        ok = false;
        return false;
    }
    static void utf8(std::string *o, uint32_t c) {
        if (c < 0x80) o->push_back((char)c);
        else if (c < 0x800) { o->push_back((char)(0xC0 | (c >> 6))); o->push_back((char)(0x80 | (c & 0x3F))); }
        else if (c < 0x10000) { o->push_back((char)(0xE0 | (c >> 12))); o->push_back((char)(0x80 | ((c >> 6) & 0x3F))); o->push_back((char)(0x80 | (c & 0x3F))); }
        else { o->push_back((char)(0xF0 | (c >> 18))); o->push_back((char)(0x80 | ((c >> 12) & 0x3F))); o->push_back((char)(0x80 | ((c >> 6) & 0x3F))); o->push_back((char)(0x80 | (c & 0x3F))); }
    }
    bool hex4(uint32_t *v) {
        if (e - p < 4) return fail();
        uint32_t x = 0;
        for (int i = 0; i < 4; i++) {
            const char c = p[i];
            x <<= 4;
            if (c >= '0' && c <= '9') x |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') x |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') x |= (uint32_t)(c - 'A' + 10);
            else return fail();
        }
        p += 4;
        *v = x;
        return true;
    }
    // at '"'; decodes into out when not null
    bool string(std::string *out) {
        if (p >= e || *p != '"') return fail();
        p++;
        for (;;) {
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
            // sixteen bytes at a time, straight to the first '"', '\\' or control character (keys and values of a kube response are 5-40 bytes: one or two steps)
            while (e - p >= 16) {
                const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p));
                const __m128i special = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(c, _mm_set1_epi8('"')), _mm_cmpeq_epi8(c, _mm_set1_epi8('\\'))),
                                                     _mm_cmpeq_epi8(_mm_max_epu8(c, _mm_set1_epi8(0x1F)), _mm_set1_epi8(0x1F)));  // (unsigned c <= 0x1F)
                const int m = _mm_movemask_epi8(special);
                const int k = m ? __builtin_ctz((unsigned)m) : 16;
                if (out && k) out->append(p, (size_t)k);
                p += k;
                if (m) break;
            }
#endif
            // eight bytes at a time while none of them is '"', '\\' or a control character (most of a kube response is string content)
            while (e - p >= 8) {
                uint64_t w;
                std::memcpy(&w, p, 8);
                const uint64_t q = w ^ 0x2222222222222222ull, b = w ^ 0x5C5C5C5C5C5C5C5Cull;
                const uint64_t hit = ((q - 0x0101010101010101ull) & ~q) | ((b - 0x0101010101010101ull) & ~b) | ((w - 0x2020202020202020ull) & ~w);
                if (hit & 0x8080808080808080ull) break;  // (may over-report above a true hit: the byte loop below decides)
                if (out) out->append(p, 8);
                p += 8;
            }
            while (p < e && *p != '"' && *p != '\\' && (unsigned char)*p >= 0x20) {  // (the rest of a window with a hit, or the last < 8 bytes)
                if (out) out->push_back(*p);
                p++;
            }
            if (!(p < e && *p != '"')) break;
            if ((unsigned char)*p < 0x20) return fail();
            if (++p >= e) return fail();  // (*p was the backslash)
            const char c = *p++;
            uint32_t u = 0;
            switch (c) {
                case '"': case '\\': case '/': if (out) out->push_back(c); break;
                case 'b': if (out) out->push_back('\b'); break;
                case 'f': if (out) out->push_back('\f'); break;
                case 'n': if (out) out->push_back('\n'); break;
                case 'r': if (out) out->push_back('\r'); break;
                case 't': if (out) out->push_back('\t'); break;
                case 'u':
                    if (!hex4(&u)) return false;
                    if (u >= 0xD800 && u < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {  // surrogate pair
                        const char *save = p;
                        p += 2;
                        uint32_t lo = 0;
                        if (!hex4(&lo)) return false;
                        if (lo >= 0xDC00 && lo < 0xE000) u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00);
                        else { p = save; u = 0xFFFD; }
                    } else if (u >= 0xD800 && u < 0xE000) u = 0xFFFD;  // lone surrogate: Go substitutes U+FFFD
                    if (out) utf8(out, u);
                    break;
                default: return fail();
            }
        }
        if (p >= e) return fail();
        p++;
        return true;
    }
    bool literal(const char *w) {
        const size_t n = std::strlen(w);
        if ((size_t)(e - p) < n || std::memcmp(p, w, n) != 0) return fail();
        p += n;
        return true;
    }
    // -? (0 | [1-9][0-9]*) (\.[0-9]+)? ([eE][+-]?[0-9]+)?  -- what encoding/json accepts: "+1", "1.2.3", "--", "1e", "01" fail the parse
    // ('failed to parse list response', postfilter.go:21-24), they are not spliced through
    bool number() {
        auto digits = [&] {
            const char *s0 = p;
            while (p < e && *p >= '0' && *p <= '9') p++;
            return p > s0;
        };
        if (p < e && *p == '-') p++;
        if (p >= e) return fail();
        if (*p == '0') p++;
        else if (*p >= '1' && *p <= '9') digits();
        else return fail();
        if (p < e && *p == '.') {
            p++;
            if (!digits()) return fail();
        }
        if (p < e && (*p == 'e' || *p == 'E')) {
            p++;
            if (p < e && (*p == '+' || *p == '-')) p++;
            if (!digits()) return fail();
        }
        return true;
    }
    // skips any value; depth-limited like encoding/json (10000)
    bool skip(int depth = 0) {
        ws();
        if (p >= e || depth > 10000) return fail();
        switch (*p) {
            case '"': return string(nullptr);
            case '{': {
                p++;
                ws();
                if (p < e && *p == '}') { p++; return true; }
                for (;;) {
                    ws();
                    if (!string(nullptr)) return false;
                    ws();
                    if (p >= e || *p++ != ':') return fail();
                    if (!skip(depth + 1)) return false;
                    ws();
                    if (p < e && *p == ',') { p++; continue; }
                    if (p < e && *p == '}') { p++; return true; }
                    return fail();
                }
            }
            case '[': {
                p++;
                ws();
                if (p < e && *p == ']') { p++; return true; }
                for (;;) {
                    if (!skip(depth + 1)) return false;
                    ws();
                    if (p < e && *p == ',') { p++; continue; }
                    if (p < e && *p == ']') { p++; return true; }
                    return fail();
                }
            }
            case 't': return literal("true");
            case 'f': return literal("false");
            case 'n': return literal("null");
            default: return number();
        }
    }
    // at '{': calls on_key(key) positioned at the value; on_key must consume the value
    template <typename F>
    bool object(F on_key) {
        if (p >= e || *p != '{') return fail();
        p++;
        ws();
        if (p < e && *p == '}') { p++; return true; }
        for (;;) {
            ws();
            // (the key in place when it has no escape: a list of 65 536 pods has a million keys at the levels that are looked at)
            std::string decoded;
            std::string_view key;
            {
                if (p >= e || *p != '"') return fail();
                const char *q = p + 1;
                while (q < e && *q != '"' && *q != '\\' && (unsigned char)*q >= 0x20) q++;
                if (q < e && *q == '"') {
                    key = std::string_view(p + 1, (size_t)(q - p - 1));
                    p = q + 1;
                } else {
                    if (!string(&decoded)) return false;
                    key = decoded;
                }
            }
            ws();
            if (p >= e || *p++ != ':') return fail();
            ws();
            if (!on_key(key)) return false;
            ws();
            if (p < e && *p == ',') { p++; continue; }
            if (p < e && *p == '}') { p++; return true; }
            return fail();
        }
    }
};

struct Item {
    size_t b, e;  // span of the element in the body
    bool is_object = false, has_meta = false;
    std::string name, ns;
};

// the item's metadata.name / metadata.namespace when they are strings (postfilter.go:73-85); later duplicates win, as in Go's map decode
bool scan_item(Scanner &s, Item *it) {
    s.ws();
    if (s.p < s.e && *s.p == '{') {
        it->is_object = true;
        return s.object([&](std::string_view k) {
            if (k != "metadata" || s.p >= s.e || *s.p != '{') {
                if (k == "metadata") it->has_meta = false;  // a later non-object "metadata" replaces an earlier object
                return s.skip();
            }
            it->has_meta = true;
            it->name.clear();
            it->ns.clear();
            return s.object([&](std::string_view mk) {
                if ((mk == "name" || mk == "namespace") && s.p < s.e && *s.p == '"') {
                    std::string v;
                    if (!s.string(&v)) return false;
                    (mk == "name" ? it->name : it->ns) = v;
                    return true;
                }
                if (mk == "name") it->name.clear();
                if (mk == "namespace") it->ns.clear();
                return s.skip();
            });
        });
    }
    return s.skip();
}

// renders one template for one item; false = "failed to resolve" (the item then gets no check from this template: postfilter.go:92-95)
bool render(const std::string &tpl, const Item &it, const std::string &user, std::string *out) {
    out->clear();
    size_t i = 0;
    while (i < tpl.size()) {
        const size_t o = tpl.find("{{", i);
        if (o == std::string::npos) {
            out->append(tpl, i, std::string::npos);
            break;
        }
        out->append(tpl, i, o - i);
        const size_t c = tpl.find("}}", o + 2);
        if (c == std::string::npos) return false;
        std::string var = tpl.substr(o + 2, c - o - 2);
        const size_t a = var.find_first_not_of(" \t"), z = var.find_last_not_of(" \t");
        var = a == std::string::npos ? "" : var.substr(a, z - a + 1);
        if (var == "name") out->append(it.name);
        else if (var == "namespace") out->append(it.ns);
        else if (var == "namespacedName") out->append(it.ns.empty() ? it.name : it.ns + "/" + it.name);
        else if (var == "user.name") out->append(user);
        else return false;  // anything richer is the Go rule engine's business
        i = c + 2;
    }
    return true;
}

// a table row (metav1.TableRow): its "object" is the row's PartialObjectMetadata (responsefilterer.go:359-364); absent, null or not an object
// = "error decoding partial object metadata from table row"
bool scan_row(Scanner &s, Item *it, bool *decodable) {
    *decodable = false;
    s.ws();
    if (s.p >= s.e || *s.p != '{') return s.skip();
    it->is_object = true;
    return s.object([&](std::string_view k) {
        if (k != "object") return s.skip();
        s.ws();
        Item inner;
        const bool obj = s.p < s.e && *s.p == '{';
        if (!scan_item(s, &inner)) return false;
        *decodable = obj;  // (a later duplicate "object" wins, as in Go's struct decode)
        it->name = inner.name;
        it->ns = inner.ns;
        return true;
    });
}

// a big copy in pieces on all host threads (a 135 MB body: 25 ms by one thread)
void big_copy(acl_engine_t *h, char *dst, const char *src, size_t n) {
    constexpr size_t kPiece = (size_t)1 << 20;
    if (n < 4 * kPiece || !h) {
        std::memcpy(dst, src, n);
        return;
    }
    host_parallel(h, (n + kPiece - 1) / kPiece, 1, [&](size_t a, size_t b) { std::memcpy(dst + a * kPiece, src + a * kPiece, std::min(n, b * kPiece) - a * kPiece); });
}

// the body with only the kept elements of the array at [arr_open, arr_close] ('[' and ']'); `none` = what an array without survivors becomes
char *splice_kept(acl_engine_t *h, const char *body, size_t body_len, size_t arr_open, size_t arr_close, const std::vector<Item> &items, const std::vector<uint8_t> &keep,
                  const char *none, size_t *out_len, size_t *kept_out) {
    size_t kept = 0, bytes = arr_open + 1 + (body_len - arr_close) + std::strlen(none);
    std::vector<size_t> at(items.size() + 1);  // where element i goes (kept ones: behind their comma)
    size_t w0 = arr_open + 1;
    for (size_t i = 0; i < items.size(); i++) {
        at[i] = w0;
        if (keep[i]) {
            w0 += (kept ? 1 : 0) + (items[i].e - items[i].b);
            kept++;
        }
    }
    at[items.size()] = w0;
    bytes += w0;
    char *o = (char *)std::malloc(std::max<size_t>(bytes, 1) + 8), *w = o;
    if (!o) return nullptr;
    if (!kept) {
        std::memcpy(w, body, arr_open);
        w += arr_open;
        std::memcpy(w, none, std::strlen(none));
        w += std::strlen(none);
        big_copy(h, w, body + arr_close + 1, body_len - arr_close - 1);
        w += body_len - arr_close - 1;
    } else {
        std::memcpy(w, body, arr_open + 1);
        auto copy_items = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                if (!keep[i]) continue;
                char *d = o + at[i];
                if (at[i] != arr_open + 1) *d++ = ',';
                std::memcpy(d, body + items[i].b, items[i].e - items[i].b);
            }
        };
        if (w0 - arr_open >= ((size_t)4 << 20) && h) host_parallel(h, items.size(), std::max<size_t>(64, items.size() / (8 * (size_t)host_threads(h))), copy_items);
        else copy_items(0, items.size());
        w = o + w0;
        std::memcpy(w, body + arr_close, body_len - arr_close);
        w += body_len - arr_close;
    }
    *out_len = (size_t)(w - o);
    *kept_out = kept;
    return o;
}

// A template cut into the six fields of `type:id#relation@type:id[#relation]` ONCE, at the separators its LITERAL text holds (the grammar of rules.go:1053-1055
// splits at the first `:`, `#`, `@`, `:`, `#`): rendering it per item and parsing the text again gives the same six fields as long as no substituted value
// holds a separator itself -- kube names and namespaces never do; a call in which one does takes the render-and-parse path for all its items.
struct CompiledTemplate {
    struct Seg {
        int var;  // -1: literal text, 0 name, 1 namespace, 2 namespacedName, 3 user.name
        std::string lit;
    };
    std::vector<Seg> field[6];
    bool compile(const std::string &tpl) {
        std::vector<Seg> segs;
        size_t i = 0;
        while (i < tpl.size()) {
            const size_t o = tpl.find("{{", i);
            if (o == std::string::npos) {
                segs.push_back(Seg{-1, tpl.substr(i)});
                break;
            }
            if (o > i) segs.push_back(Seg{-1, tpl.substr(i, o - i)});
            const size_t c = tpl.find("}}", o + 2);
            if (c == std::string::npos) return false;
            std::string var = tpl.substr(o + 2, c - o - 2);
            const size_t a = var.find_first_not_of(" \t"), z = var.find_last_not_of(" \t");
            var = a == std::string::npos ? "" : var.substr(a, z - a + 1);
            const int v = var == "name" ? 0 : var == "namespace" ? 1 : var == "namespacedName" ? 2 : var == "user.name" ? 3 : -1;
            if (v < 0) return false;
            segs.push_back(Seg{v, std::string()});
            i = c + 2;
        }
        static const char kSep[5] = {':', '#', '@', ':', '#'};
        int f = 0;
        for (const Seg &sg : segs) {
            if (sg.var >= 0) {
                field[f].push_back(sg);
                continue;
            }
            size_t from = 0;
            while (f < 5) {
                const size_t at = sg.lit.find(kSep[f], from);
                if (at == std::string::npos) break;
                if (at > from) field[f].push_back(Seg{-1, sg.lit.substr(from, at - from)});
                from = at + 1;
                f++;
            }
            if (from < sg.lit.size()) field[f].push_back(Seg{-1, sg.lit.substr(from)});
        }
        return f >= 4;  // (f == 4: no subject relation)
    }
    bool per_item(int f) const {
        for (const Seg &sg : field[f])
            if (sg.var >= 0 && sg.var != 3) return true;
        return false;
    }
};
inline bool has_separator(std::string_view v) {
    for (char c : v)
        if (c == ':' || c == '#' || c == '@') return true;
    return false;
}

// The elements of the array that s.p stands at (`[`): their spans, their metadata (scan_item; table rows: scan_row) -- and s.p behind the array's `]`.
// Small arrays: element by element.  From kParallelBytes on: the element spans by all host threads (json_index.hpp: every chunk indexed under both
// in-string hypotheses, a sequential fix-up over the chunks), then every span scanned -- validated, its metadata taken -- in parallel.  Which bodies are JSON
// does not change: an array is valid exactly when its spans are valid values separated by single commas, and the scanner decides that per span.
constexpr size_t kParallelBytes = (size_t)1 << 19;
bool scan_array(acl_engine_t *h, Scanner &s, const char *body, bool table, std::vector<Item> *items, std::vector<uint8_t> *decodable, size_t *arr_open, size_t *arr_close,
                size_t chunk_bytes = 0, double *ms_index = nullptr) {
    const auto t_0 = std::chrono::steady_clock::now();
    items->clear();
    if (decodable) decodable->clear();
    *arr_open = (size_t)(s.p - body);
    const size_t region_b = *arr_open + 1, region_e = (size_t)(s.e - body);
    if (!chunk_bytes && (region_e - region_b < kParallelBytes || host_threads(h) <= 1)) {
        s.p++;
        s.ws();
        if (s.p < s.e && *s.p == ']') {
            *arr_close = (size_t)(s.p - body);
            s.p++;
            return true;
        }
        for (;;) {
            s.ws();
            Item it;
            bool dec = true;
            const char *b0 = s.p;
            if (!(table ? scan_row(s, &it, &dec) : scan_item(s, &it))) return false;
            it.b = (size_t)(b0 - body);
            it.e = (size_t)(s.p - body);
            if (decodable) decodable->push_back(dec && it.is_object);
            items->push_back(std::move(it));
            s.ws();
            if (s.p < s.e && *s.p == ',') { s.p++; continue; }
            if (s.p < s.e && *s.p == ']') {
                *arr_close = (size_t)(s.p - body);
                s.p++;
                return true;
            }
            return s.fail();
        }
    }
    // ---- structure: chunks of the rest of the body (the array is nearly all of it), four per thread so that a slow one does not hold the others up
    const size_t len = region_e - region_b;
    size_t chunk = chunk_bytes ? chunk_bytes : std::min<size_t>((size_t)1 << 20, std::max<size_t>((size_t)1 << 16, len / (4 * (size_t)host_threads(h))));
    chunk = (chunk + 63) & ~(size_t)63;
    const size_t nchunks = (len + chunk - 1) / chunk;
    std::vector<jsonidx::Chunk> chunks(nchunks);
    host_parallel(h, nchunks, 1, [&](size_t a, size_t b) {
        for (size_t c = a; c < b; c++) jsonidx::index_chunk(body, region_b, region_b + c * chunk, std::min(region_e, region_b + (c + 1) * chunk), &chunks[c]);
    });
    std::vector<jsonidx::Span> spans;
    if (!jsonidx::array_spans(chunks, region_b, &spans, arr_close)) return s.fail();
    chunks.clear();
    if (ms_index) *ms_index = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count();
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; };
    for (jsonidx::Span &sp : spans) {
        while (sp.b < sp.e && is_ws(body[sp.b])) sp.b++;
        while (sp.e > sp.b && is_ws(body[sp.e - 1])) sp.e--;
    }
    if (spans.size() == 1 && spans[0].b == spans[0].e) spans.clear();  // `[ ]`
    // ---- every element through the scanner (an empty span -- `[1,,2]` -- fails there like any other non-value)
    items->resize(spans.size());
    if (decodable) decodable->resize(spans.size());
    std::atomic<bool> bad{false};
    host_parallel(h, spans.size(), std::max<size_t>(16, spans.size() / (8 * (size_t)host_threads(h))), [&](size_t a, size_t b) {
        for (size_t i = a; i < b && !bad.load(std::memory_order_relaxed); i++) {
            Scanner es{body + spans[i].b, body + spans[i].e};
            Item &it = (*items)[i];
            bool dec = true;
            const bool ok = table ? scan_row(es, &it, &dec) : scan_item(es, &it);
            if (!ok || !es.ok || es.p != es.e) {
                bad.store(true, std::memory_order_relaxed);
                return;
            }
            it.b = spans[i].b;
            it.e = spans[i].e;
            if (decodable) (*decodable)[i] = dec && it.is_object;
        }
    });
    if (bad.load()) return s.fail();
    s.p = body + *arr_close + 1;
    return true;
}

}  // namespace

extern "C" {

int acl_filter_list_response(acl_engine_t *h, const char *body, size_t body_len, const char *const *templates, size_t n_templates, const char *user_name,
                             char **out_body, size_t *out_len, uint64_t *kept_out, uint64_t *total_out) {
    return acl_filter_list_response_req(h, body, body_len, templates, n_templates, user_name, nullptr, out_body, out_len, kept_out, total_out);
}

int acl_filter_list_response_req(acl_engine_t *h, const char *body, size_t body_len, const char *const *templates, size_t n_templates, const char *user_name,
                                 const acl_list_request_t *req, char **out_body, size_t *out_len, uint64_t *kept_out, uint64_t *total_out) {
    if (!body || !out_body || !out_len || (n_templates && !templates)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_filter_list_response: NULL argument");
    *out_body = nullptr;
    *out_len = 0;
    auto unchanged = [&](uint64_t n) {
        char *o = (char *)std::malloc(std::max<size_t>(body_len, 1));
        if (!o) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "out of host memory");
        big_copy(h, o, body, body_len);
        *out_body = o;
        *out_len = body_len;
        if (kept_out) *kept_out = n;
        if (total_out) *total_out = n;
        return (int)ACL_OK;
    };
    static const bool kTrace = getenv("ACL_DEBUG_LIST") != nullptr;  // (phase times of the call on stderr)
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count(); };
    double ms_scan = 0, ms_resolve = 0, ms_keep = 0, ms_index = 0;
    // ---- one scan: the top-level object's "items" array and its elements
    Scanner s{body, body + body_len};
    s.ws();
    std::vector<Item> items;
    bool have_items = false;
    size_t arr_open = 0, arr_close = 0;  // positions of '[' and ']'
    if (s.p >= s.e || *s.p != '{') return fail(ACL_ERR_INVALID_ARGUMENT, "failed to parse list response: not a JSON object");
    const bool parsed = s.object([&](std::string_view k) {
        if (k != "items" || s.p >= s.e || *s.p != '[') {
            if (k == "items") have_items = false;
            return s.skip();
        }
        have_items = true;
        return scan_array(h, s, body, false, &items, nullptr, &arr_open, &arr_close, 0, &ms_index);
    });
    s.ws();
    if (!parsed || !s.ok || s.p != s.e) return fail(ACL_ERR_INVALID_ARGUMENT, "failed to parse list response: invalid JSON");
    if (!have_items || items.empty()) return unchanged(0);  // postfilter.go:26-35: nothing to filter, the body stays as it is
    ms_scan = ms_since();
    // ---- resolve K x F pairs
    const std::string user = user_name ? user_name : "";
    std::vector<std::string> tpls(templates, templates + n_templates);
    const size_t F = tpls.size();
    std::vector<uint32_t> off(items.size() + 1, 0);
    std::vector<acl_check_item_v_t> ci;
    // every item's template input is rules.NewResolveInput(input.Request, ...) (postfilter.go:88, rules.go:315-342): the item's own metadata first,
    // the REQUEST's name / namespace where the item has none, and no namespace at all for the `namespaces` resource (its requests carry the
    // namespace name in both fields)
    const std::string req_name = req && req->name ? req->name : "", req_ns = req && req->namespace_ ? req->namespace_ : "";
    const bool cluster_scoped = req && req->resource && std::strcmp(req->resource, "namespaces") == 0;
    // ---- the usual call: every template cut into its six fields once, the per-item fields rendered straight into the views' bytes
    std::vector<CompiledTemplate> ct(F);
    std::vector<std::array<std::string, 6>> cst(F);  // the fields that do not depend on the item
    std::vector<std::unique_ptr<char[]>> arenas;
    std::mutex arenas_mu;
    std::vector<std::array<bool, 6>> varies(F);      // ... and the ones that do
    bool compiled = F > 0 && !has_separator(user);
    for (size_t t = 0; t < F && compiled; t++) {
        compiled = ct[t].compile(tpls[t]);
        for (int f = 0; f < 6 && compiled; f++) {
            varies[t][f] = ct[t].per_item(f);
            if (!varies[t][f])
                for (const auto &sg : ct[t].field[f]) cst[t][f] += sg.var == 3 ? user : sg.lit;
        }
    }
    if (compiled) {
        for (size_t i = 0; i < items.size(); i++) off[i + 1] = off[i] + (items[i].is_object ? (uint32_t)F : 0u);
        ci.resize(off[items.size()]);
        std::atomic<bool> odd{false};
        host_parallel(h, items.size(), std::max<size_t>(256, items.size() / (8 * (size_t)host_threads(h))), [&](size_t a, size_t b) {
            static const std::string kNone;
            auto value = [&](const Item &it, int var, char *w) -> size_t {  // bytes of a variable (written at w when not null)
                const std::string *nm = it.name.empty() ? &req_name : &it.name, *ns = cluster_scoped ? &kNone : (it.ns.empty() ? &req_ns : &it.ns);
                auto put = [&](const std::string &x, size_t at) {
                    if (w) std::memcpy(w + at, x.data(), x.size());
                    return x.size();
                };
                if (var == 0) return put(*nm, 0);
                if (var == 1) return put(*ns, 0);
                if (var == 3) return put(user, 0);
                if (ns->empty()) return put(*nm, 0);
                const size_t n = put(*ns, 0);
                if (w) w[n] = '/';
                return n + 1 + put(*nm, n + 1);
            };
            size_t bytes = 0;
            for (size_t i = a; i < b; i++) {
                if (!items[i].is_object) continue;  // postfilter.go:68-71
                const std::string &nm = items[i].name.empty() ? req_name : items[i].name, &ns = items[i].ns.empty() ? req_ns : items[i].ns;
                if (has_separator(nm) || (!cluster_scoped && has_separator(ns))) {
                    odd.store(true, std::memory_order_relaxed);
                    return;
                }
                for (size_t t = 0; t < F; t++)
                    for (int f = 0; f < 6; f++)
                        if (varies[t][f])
                            for (const auto &sg : ct[t].field[f]) bytes += sg.var < 0 ? sg.lit.size() : value(items[i], sg.var, nullptr);
            }
            std::unique_ptr<char[]> arena(new char[std::max<size_t>(bytes, 1)]);
            char *w = arena.get();
            for (size_t i = a; i < b; i++) {
                if (!items[i].is_object) continue;
                for (size_t t = 0; t < F; t++) {
                    acl_str_t fld[6];
                    for (int f = 0; f < 6; f++) {
                        if (!varies[t][f]) {
                            fld[f] = acl_str_t{cst[t][f].data(), cst[t][f].size()};
                            continue;
                        }
                        char *const f0 = w;
                        for (const auto &sg : ct[t].field[f]) {
                            if (sg.var < 0) {
                                std::memcpy(w, sg.lit.data(), sg.lit.size());
                                w += sg.lit.size();
                            } else w += value(items[i], sg.var, w);
                        }
                        fld[f] = acl_str_t{f0, (size_t)(w - f0)};
                    }
                    if (fld[5].n == 0) fld[5] = acl_str_t{nullptr, 0};
                    ci[off[i] + t] = acl_check_item_v_t{fld[0], fld[1], fld[2], fld[3], fld[4], fld[5]};
                }
            }
            std::lock_guard<std::mutex> lk(arenas_mu);
            arenas.push_back(std::move(arena));
        });
        if (odd.load()) compiled = false;  // (a name with a separator in it: the text decides, for every item)
    }
    std::vector<RelText> rels;
    if (!compiled) {
    rels.resize(items.size() * F);
    std::vector<uint32_t> cnt(items.size(), 0);  // item i's resolved pairs: rels[i F ...], cnt[i] of them
    host_parallel(h, items.size(), std::max<size_t>(256, items.size() / (8 * (size_t)host_threads(h))), [&](size_t a, size_t b) {
        std::string text;
        for (size_t i = a; i < b; i++) {
            if (!items[i].is_object) continue;  // postfilter.go:68-71
            if (items[i].name.empty()) items[i].name = req_name;
            if (items[i].ns.empty()) items[i].ns = req_ns;
            if (cluster_scoped) items[i].ns.clear();
            for (const std::string &t : tpls) {
                RelText &r = rels[i * F + cnt[i]];
                if (!render(t, items[i], user, &text) || !parse_relationship_text(text, &r)) continue;  // resolution failed: no check (postfilter.go:92-95)
                cnt[i]++;
            }
        }
    });
    for (size_t i = 0; i < items.size(); i++) off[i + 1] = off[i] + cnt[i];
    const size_t npairs = off[items.size()];
    if (!npairs) return unchanged(items.size());  // postfilter.go:122-125
    // {pointer, length} views through acl_check_bulk_keep_v: a list filtered for ONE user by one template -- every pair shares type, permission and subject --
    // is answered by one reverse walk and K bit tests (engine.cpp keep_by_reverse_walk), any other shape by the forward path
    ci.assign(npairs, acl_check_item_v_t{});
    host_parallel(h, items.size(), std::max<size_t>(1024, items.size() / (4 * (size_t)host_threads(h))), [&](size_t a, size_t b) {
        auto sv = [](const std::string &x) { return acl_str_t{x.data(), x.size()}; };
        for (size_t i = a; i < b; i++)
            for (uint32_t j = 0; j < cnt[i]; j++) {
                const RelText &r = rels[i * F + j];
                acl_check_item_v_t &c = ci[off[i] + j];
                c = acl_check_item_v_t{sv(r.rtype), sv(r.rid), sv(r.rel), sv(r.stype), sv(r.sid), sv(r.srel)};
                if (r.srel.empty()) c.subject_relation = acl_str_t{nullptr, 0};
            }
    });
    }
    if (ci.empty()) return unchanged(items.size());  // postfilter.go:122-125
    ms_resolve = ms_since();
    std::vector<uint8_t> keep(items.size());
    int rc = acl_check_bulk_keep_v(h, ci.data(), ci.size(), off.data(), items.size(), keep.data());
    if (rc) return rc;
    ms_keep = ms_since();
    // ---- splice: the original bytes minus the dropped items (the reference appends to a nil slice, postfilter.go:142: with nothing allowed, "items" marshals as null)
    size_t kept = 0;
    char *o = splice_kept(h, body, body_len, arr_open, arr_close, items, keep, "null", out_len, &kept);
    if (!o) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "out of host memory");
    *out_body = o;
    if (kept_out) *kept_out = kept;
    if (total_out) *total_out = items.size();
    if (kTrace) std::fprintf(stderr, "list filter: %zu items, %.1f MB | element spans at %.2f ms | elements scanned at %.2f | pairs resolved at %.2f | kept known at %.2f | spliced at %.2f (%zu kept)\n", items.size(), body_len / 1e6, ms_index, ms_scan, ms_resolve, ms_keep, ms_since(), kept);
    return ACL_OK;
}

int acl_prefilter_response(acl_engine_t *h, int type, const uint32_t *bitmap, size_t bitmap_words, const char *id_template, int kind, const char *body, size_t body_len,
                           char **out_body, size_t *out_len, uint64_t *kept_out, uint64_t *total_out) {
    if (!body || !out_body || !out_len || !id_template || (bitmap_words && !bitmap) || kind < ACL_BODY_LIST || kind > ACL_BODY_OBJECT)
        return fail(ACL_ERR_INVALID_ARGUMENT, "acl_prefilter_response: bad argument");
    *out_body = nullptr;
    *out_len = 0;
    const std::string tpl = id_template;
    auto unchanged = [&](uint64_t kept, uint64_t total) {
        char *o = (char *)std::malloc(std::max<size_t>(body_len, 1));
        if (!o) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "out of host memory");
        big_copy(h, o, body, body_len);
        *out_body = o;
        *out_len = body_len;
        if (kept_out) *kept_out = kept;
        if (total_out) *total_out = total;
        return (int)ACL_OK;
    };
    Scanner s{body, body + body_len};
    s.ws();
    if (s.p >= s.e || *s.p != '{') return fail(ACL_ERR_INVALID_ARGUMENT, "failed to decode response body: not a JSON object");
    std::vector<Item> items;
    std::vector<uint8_t> decodable;  // tables: the row's object could be decoded
    bool have = false;
    size_t arr_open = 0, arr_close = 0;
    Item single;
    bool parsed;
    if (kind == ACL_BODY_OBJECT) {
        parsed = scan_item(s, &single);
    } else {
        const char *const key = kind == ACL_BODY_TABLE ? "rows" : "items";
        parsed = s.object([&](std::string_view k) {
            if (k != key || s.p >= s.e || *s.p != '[') {
                if (k == key) have = false;
                return s.skip();
            }
            have = true;
            return scan_array(h, s, body, kind == ACL_BODY_TABLE, &items, &decodable, &arr_open, &arr_close);
        });
    }
    s.ws();
    if (!parsed || !s.ok || s.p != s.e) return fail(ACL_ERR_INVALID_ARGUMENT, "failed to decode response body: invalid JSON");
    // ---- IsAllowed(namespace, name): the object id the rule maps this (namespace, name) to, looked up, its bit tested
    std::shared_lock<std::shared_mutex> nlk(h->names_mu);
    const Schema &sc = h->store.schema();
    if (type < 0 || type >= (int)sc.defs.size()) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_prefilter_response: unknown object type");
    const ObjectTable &ot = h->store.objects(type);
    std::string idtext;
    auto allowed = [&](const Item &it) {
        uint32_t id;
        return render(tpl, it, std::string(), &idtext) && ot.find(idtext, &id) && (size_t)(id >> 5) < bitmap_words && ((bitmap[id >> 5] >> (id & 31u)) & 1u);
    };
    if (kind == ACL_BODY_OBJECT) {  // responsefilterer.go:401-416: the object itself, or "unauthorized" (writeResp turns that into a 401 body)
        if (!allowed(single)) return fail(ACL_ERR_PERMISSION_DENIED, "unauthorized");
        return unchanged(1, 1);
    }
    if (!have) return unchanged(0, 0);  // (nothing to cut; the reference's re-encode would add an empty array)
    for (size_t i = 0; i < items.size(); i++)
        if (!decodable[i])
            return fail(ACL_ERR_INVALID_ARGUMENT, kind == ACL_BODY_TABLE ? "error decoding partial object metadata from table row" : "failed to decode response body: list item is not an object");
    std::vector<uint8_t> keep(items.size());
    // (names_mu shared, then the pool: the order of the interning callers -- InternPool::run)
    host_parallel(h, items.size(), std::max<size_t>(256, items.size() / (8 * (size_t)host_threads(h))), [&](size_t a, size_t b) {
        std::string idt;
        for (size_t i = a; i < b; i++) {
            uint32_t id;
            keep[i] = render(tpl, items[i], std::string(), &idt) && ot.find(idt, &id) && (size_t)(id >> 5) < bitmap_words && ((bitmap[id >> 5] >> (id & 31u)) & 1u);
        }
    });
    nlk.unlock();
    // both consumers start from make([]T, 0): an array without survivors is [], not null (responsefilterer.go:358,375)
    size_t kept = 0;
    char *o = items.empty() ? nullptr : splice_kept(h, body, body_len, arr_open, arr_close, items, keep, "[]", out_len, &kept);
    if (items.empty()) return unchanged(0, 0);
    if (!o) return fail(ACL_ERR_RESOURCE_EXHAUSTED, "out of host memory");
    *out_body = o;
    if (kept_out) *kept_out = kept;
    if (total_out) *total_out = items.size();
    return ACL_OK;
}

// Test hook: the elements of the array at body[arr_open] (`[`) as the list filters find them -- chunk_bytes > 0: through the parallel index with chunks of
// that size (tests/test_list_filter.py walks it over random documents with chunks of 64 bytes and up), 0: as a call of that size would.  spans_out: {begin, end}
// per element (at most cap elements written), *n_out: elements found, *close_out: the `]`.  Touches no device and no store: any engine will do.
int acl_selfcheck_json_array(acl_engine_t *h, const char *body, size_t body_len, size_t arr_open, size_t chunk_bytes, size_t *spans_out, size_t cap, size_t *n_out, size_t *close_out) {
    if (!body || arr_open >= body_len || body[arr_open] != '[' || !n_out || !close_out || (cap && !spans_out)) return fail(ACL_ERR_INVALID_ARGUMENT, "acl_selfcheck_json_array: bad argument");
    Scanner s{body + arr_open, body + body_len};
    std::vector<Item> items;
    size_t ao = 0, ac = 0;
    if (!scan_array(h, s, body, false, &items, nullptr, &ao, &ac, chunk_bytes) || !s.ok) return fail(ACL_ERR_INVALID_ARGUMENT, "invalid JSON");
    for (size_t i = 0; i < items.size() && i < cap; i++) {
        spans_out[2 * i] = items[i].b;
        spans_out[2 * i + 1] = items[i].e;
    }
    *n_out = items.size();
    *close_out = ac;
    return ACL_OK;
}

}  // extern "C"
