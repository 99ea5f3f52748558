// name_probe.hpp -- what k_resolve_names (kernels.hip) does with ONE 64-byte name record, written once: the kernel instantiates it with the
// device's 64 x 64 -> high-half multiply, acl_selfcheck_names (engine.cpp) with the host's and runs it on the CPU over a byte copy of the slot
// arrays that is kept current exactly as the HBM mirror is (engine_names.cpp) -- so the record layout, the hash, the probe and the change lists
// are covered by the tests that run without a GPU (tests/test_store_cpu.py::test_names_resolved_over_a_copy_of_the_tables).
#pragma once
#include <cstdint>

#include "kernels.hpp"

namespace acl {

struct HostMulHi {
    uint64_t operator()(uint64_t a, uint64_t b) const { return (uint64_t)(((__uint128_t)a * b) >> 64); }
};

// ObjectTable::hash (store.cpp), bit for bit, over a name held as little-endian dwords (whatever lies behind the name's last byte is masked off)
template <class MulHi>
__host__ __device__ __forceinline__ uint64_t name_hash(const uint32_t *w, uint32_t n, MulHi mulhi) {
    auto fold = [&](uint64_t a, uint64_t b) { return (a * b) ^ mulhi(a, b); };
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n * 0xD6E8FEB86659FD93ull);
    uint32_t k = 0, left = n;
    for (; left >= 8; left -= 8, k += 2) h = fold(h ^ ((uint64_t)w[k] | ((uint64_t)w[k + 1] << 32)), 0xE7037ED1A0B428DBull);
    if (left) {
        uint32_t lo = w[k], hi = left > 4 ? w[k + 1] : 0u;
        if (left < 4) lo &= (1u << (8 * left)) - 1u;
        else if (left > 4) hi &= (1u << (8 * (left - 4))) - 1u;
        h = fold(h ^ ((uint64_t)lo | ((uint64_t)hi << 32)), 0x8EBC6AF09C88C6E3ull);
    }
    return fold(h, 0x589965CC75374CC3ull) ^ h;
}

// ObjectTable::find_hashed over the raw slot array, for names of at most 46 bytes (longer ones never reach the device: engine.cpp pack_names)
template <class MulHi>
__host__ __device__ __forceinline__ bool name_find(const NameTab &t, const uint32_t *w, uint32_t n, uint32_t *id_out, MulHi mulhi) {
    if (!t.slots || !t.cap) return false;
    const uint64_t h = name_hash(w, n, mulhi);
    const uint32_t tag = (uint32_t)(h >> 32), nd = (n + 3) >> 2;
    const uint32_t last_mask = (n & 3u) ? (1u << (8 * (n & 3u))) - 1u : 0xFFFFFFFFu;
    uint32_t i = (uint32_t)(((h & 0xFFFFFFFFull) * t.cap) >> 32);
    for (uint32_t step = 0; step < t.cap; step++, i = (i + 1 == t.cap) ? 0u : i + 1) {
        const uint4 *sl = t.slots + (size_t)i * 4;
        const uint4 a = sl[0];  // tag, id, length | the name's first two bytes, its next four
        if (a.y == 0xFFFFFFFFu) return false;  // an empty slot ends the probe
        if (a.y == 0xFFFFFFFEu || a.x != tag || (a.z & 0xFFFFu) != n) continue;  // (tombstones are walked over)
        const uint4 b = sl[1], c = sl[2], d = sl[3];
        const uint32_t sw[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        bool eq = true;
#pragma unroll
        for (uint32_t j = 0; j < 12; j++) {  // the name starts at byte 10 of the slot: dword j of it straddles slot dwords 2 + j and 3 + j
            const uint32_t have = (sw[2 + j] >> 16) | (sw[3 + j] << 16);
            if (j < nd) eq = eq && (((have ^ w[j]) & (j + 1 == nd ? last_mask : 0xFFFFFFFFu)) == 0u);
        }
        if (eq) {
            *id_out = a.y;
            return true;
        }
    }
    return false;
}

// One record (its 16 dwords at w) -> the item; *unknown: one of its names is in no table
template <class MulHi>
__host__ __device__ __forceinline__ uint4 name_resolve_record(const NameTab *tabs, const uint32_t *w, bool *unknown, MulHi mulhi) {
    const uint32_t rt = w[0] & 0xFFFFu, st = w[1] & 0xFFFFu, rlen = w[2] & 0xFFu, slen = (w[2] >> 8) & 0xFFu;
    *unknown = false;
    if (rt == 0xFFFFu) return make_uint4(0xFFFFu, 0u, 0xFFFFu, 0u);  // the host marked the item: unknown type / permission, an empty or ill-formed field
    const uint32_t *rw = w + 3, *sw = w + 3 + ((rlen + 3) >> 2);
    uint32_t res = 0, sub = 0;
    const bool kr = name_find(tabs[rt], rw, rlen, &res, mulhi), ks = name_find(tabs[st], sw, slen, &sub, mulhi);
    if (!kr || !ks) {
        bool same = !kr && !ks && rt == st && rlen == slen;
        const uint32_t nd = (rlen + 3) >> 2;
        for (uint32_t j = 0; same && j < nd; j++) same = rw[j] == sw[j];  // (both zero-padded by the host)
        if (same) res = sub = kUnknownSame;
        else {
            if (!kr) res = kUnknownRes;
            if (!ks) sub = kUnknownSub;
        }
        *unknown = true;
    }
    return make_uint4(w[0], res, w[1], sub);
}

}  // namespace acl
