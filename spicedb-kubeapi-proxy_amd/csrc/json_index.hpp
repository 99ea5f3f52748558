// json_index.hpp -- where the elements of a big JSON array begin and end, found by all host threads at once.
//
// The list-level filters (engine_list.cpp: filterListResponse, pkg/authz/postfilter.go:17-55; filterList / filterTable, responsefilterer.go:349-400) cut
// items out of a kube list response that can be a hundred megabytes.  The K checks behind them are 0.1-0.2 ms on the device; scanning the body for the item
// spans with one thread was 80 ms for 65 536 pods (1.7 GB/s).  A JSON document cannot be split at arbitrary offsets -- whether a brace is structure or string
// content depends on every quote before it -- but that dependence is ONE BIT per chunk: "does the chunk begin inside a string".  So every chunk is indexed
// under BOTH answers in one pass (the in-string mask of one hypothesis is the complement of the other's), and a short sequential fix-up picks, chunk by chunk,
// the hypothesis the previous chunk's quote parity implies, adds up the bracket depths and keeps the commas that separate the array's elements.
// 64 bytes per step: SSE2 compares -> bit masks of quotes, backslashes, brackets and commas; escaped characters by the odd-backslash-run carry trick; the
// in-string mask by a prefix XOR over the unescaped quotes.  Nothing here VALIDATES: the caller runs its scanner over every element span it is given (in
// parallel) and over the document around the array, and that, not this index, decides whether the body is JSON -- spans that tile the array, are separated
// by single commas and are each a valid value ARE a valid array, however they were found.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

namespace jsonidx {

struct Masks {
    uint64_t quote, bslash, open, close, comma;
};

// 64 bytes at p (all readable)
static inline Masks masks64(const char *p) {
    Masks m{0, 0, 0, 0, 0};
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
    const __m128i vq = _mm_set1_epi8('"'), vb = _mm_set1_epi8('\\'), vo = _mm_set1_epi8('{'), vc = _mm_set1_epi8('}'), vm = _mm_set1_epi8(','), v20 = _mm_set1_epi8(0x20);
    for (int k = 0; k < 4; k++) {
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 16 * k));
        const __m128i l = _mm_or_si128(c, v20);  // '[' | 0x20 == '{', ']' | 0x20 == '}' (and nothing else maps there: 0x5B / 0x5D / 0x7B / 0x7D only)
        m.quote |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(c, vq)) << (16 * k);
        m.bslash |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(c, vb)) << (16 * k);
        m.open |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(l, vo)) << (16 * k);
        m.close |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(l, vc)) << (16 * k);
        m.comma |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(c, vm)) << (16 * k);
    }
#else
    for (int i = 0; i < 64; i++) {
        const char c = p[i];
        const uint64_t b = 1ull << i;
        if (c == '"') m.quote |= b;
        else if (c == '\\') m.bslash |= b;
        else if (c == '{' || c == '[') m.open |= b;
        else if (c == '}' || c == ']') m.close |= b;
        else if (c == ',') m.comma |= b;
    }
#endif
    return m;
}

// the characters a backslash escapes (bit i: byte i is preceded by an odd run of backslashes); `carry`: the block's first byte is escaped / the next block's is.
// A run that starts on an odd bit and has odd length ends on ... -- the add ripples each run to its end, where the parity of the start decides.
static inline uint64_t escaped_mask(uint64_t bs, uint64_t *carry) {
    if (!bs) {
        const uint64_t e = *carry;
        *carry = 0;
        return e;
    }
    bs &= ~*carry;  // (an escaped backslash starts no run)
    const uint64_t follows = (bs << 1) | *carry;
    const uint64_t even = 0x5555555555555555ull;
    const uint64_t odd_starts = bs & ~even & ~follows;
    uint64_t even_seq;
    *carry = __builtin_add_overflow(odd_starts, bs, &even_seq) ? 1u : 0u;
    const uint64_t invert = even_seq << 1;
    return (even ^ invert) & follows;
}

static inline uint64_t prefix_xor(uint64_t x) {
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    x ^= x << 32;
    return x;
}

struct Event {
    size_t pos;
    int32_t rel;  // bracket depth relative to the chunk's first byte (0 there)
};
// What a chunk looks like under one hypothesis about its first byte (outside / inside a string)
struct Track {
    int32_t depth = 0, minv = 0;
    std::vector<Event> commas;  // commas at rel <= 0: the array's separators are the ones at rel == -(depth at the chunk's start)
    std::vector<Event> mins;    // every closing bracket that takes the depth below everything before it in the chunk: the array's `]` is one of them
    inline void walk(size_t base, uint64_t open, uint64_t close, uint64_t comma) {
        uint64_t all = open | close | comma;
        while (all) {
            const int i = __builtin_ctzll(all);
            const uint64_t b = all & (0 - all);
            all ^= b;
            if (open & b) depth++;
            else if (close & b) {
                depth--;
                if (depth < minv) {
                    minv = depth;
                    mins.push_back(Event{base + (size_t)i, depth});
                }
            } else if (depth <= 0) commas.push_back(Event{base + (size_t)i, depth});
        }
    }
};
struct Chunk {
    Track t[2];        // [0]: the chunk begins outside a string, [1]: inside one
    bool flips = false;  // an odd number of unescaped quotes: the next chunk begins in the other state
};

// indexes [b, e) of `body`; region_b: where the indexed region begins (the look-back for a backslash run stops there)
static inline void index_chunk(const char *body, size_t region_b, size_t b, size_t e, Chunk *out) {
    uint64_t esc = 0;
    {   // is the first byte escaped?  (an odd run of backslashes right before it)
        size_t k = b;
        while (k > region_b && body[k - 1] == '\\') k--;
        esc = (b - k) & 1u;
    }
    uint64_t in0 = 0;  // all-ones while inside a string, under hypothesis 0
    bool odd = false;
    char tail[64];
    for (size_t p = b; p < e; p += 64) {
        const char *src = body + p;
        uint64_t valid = ~0ull;
        if (e - p < 64) {
            std::memset(tail, ' ', 64);
            std::memcpy(tail, src, e - p);
            src = tail;
            valid = (1ull << (e - p)) - 1;
        }
        Masks m = masks64(src);
        const uint64_t escd = escaped_mask(m.bslash & valid, &esc);
        const uint64_t q = m.quote & ~escd & valid;
        const uint64_t in = prefix_xor(q) ^ in0;  // (bit i set: byte i is inside a string -- or is its opening quote -- under hypothesis 0)
        in0 = (uint64_t)((int64_t)in >> 63);
        odd ^= (__builtin_popcountll(q) & 1) != 0;
        const uint64_t st = (m.open | m.close | m.comma) & valid;
        if (st & ~in) out->t[0].walk(p, m.open & ~in, m.close & ~in, m.comma & ~in);
        if (st & in) out->t[1].walk(p, m.open & in, m.close & in, m.comma & in);
    }
    out->flips = odd;
}

struct Span {
    size_t b, e;
};

// The fix-up over the chunks of the region that begins right after the array's `[` (outside a string, depth 0): element separators and the closing bracket.
// false: the region ends before the array does.
static inline bool array_spans(const std::vector<Chunk> &chunks, size_t region_b, std::vector<Span> *spans, size_t *arr_close) {
    spans->clear();
    int64_t depth = 0;
    int hyp = 0;
    size_t from = region_b;
    for (const Chunk &c : chunks) {
        const Track &t = c.t[hyp];
        const int64_t want = -depth;
        size_t close_at = (size_t)-1;
        for (const Event &m : t.mins)
            if (m.rel == want - 1) {
                close_at = m.pos;
                break;
            }
        for (const Event &cm : t.commas) {
            if (cm.pos > close_at) break;
            if (cm.rel == want) {
                spans->push_back(Span{from, cm.pos});
                from = cm.pos + 1;
            }
        }
        if (close_at != (size_t)-1) {
            spans->push_back(Span{from, close_at});
            *arr_close = close_at;
            return true;
        }
        depth += t.depth;
        if (c.flips) hyp ^= 1;
    }
    return false;
}

}  // namespace jsonidx
