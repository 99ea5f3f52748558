// kernels.hpp -- launch interface of the gfx950 frontier-expansion kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.hpp"

namespace acl {

// frontier geometry
constexpr uint32_t kChunk = 1024;         // entries per frontier chunk (16 KiB)
constexpr uint32_t kSegsPerChunk = kChunk / 64;
constexpr uint32_t kMaxLevels = 50;       // dispatch max depth, reference pkg/spicedb/spicedb.go:34
constexpr uint32_t kLevelSlots = 64;      // per-iteration chunk counters
constexpr uint32_t kDeadMeta = 0xFFFFFFFFu;

// per-item status byte written by the kernels
enum : uint8_t { ITEM_ERR_NONE = 0, ITEM_ERR_DEPTH = 1, ITEM_ERR_INVALID = 2 };

// frontier entry: one pending sub-check (req, state) -- 16 B, one dwordx4 per lane
//   x = object id, y = request index, z = meta, w = subject id of the request
//   meta = slot[0:13) | level[13:19) | subject key[19:32)
struct DevGraph {
    const uint32_t *off, *edges;
    const FwdOp *ops;
    const SlotProg *progs;
    const uint32_t *type_slot_base, *type_nmembers;
    uint32_t nslots, ntypes;
};
struct DevReverse {
    const uint32_t *roff, *redges;
    const RevOp *rops;
    const RevProg *rprogs, *rseeds;
    const uint32_t *slot_bit_base;  // [nslots]
    const uint32_t *slot_nobjects;  // [nslots] id space of the slot's type
    uint32_t *visited;              // [nreq][visited_words]
    uint32_t visited_words;
};
struct DevFrontier {
    uint4 *buf[2];
    uint32_t *counts[2];   // per-chunk fill
    uint32_t *nchunks;     // [kLevelSlots] chunks produced by iteration i (0 = seed)
    uint32_t *overflow;    // set when a frontier buffer ran out of chunks
    uint32_t max_chunks;
};

void launch_seed(hipStream_t s, const DevGraph &g, const DevFrontier &f, const uint4 *items, uint32_t n, uint32_t req_base, uint8_t *has, uint8_t *err);
void launch_expand(hipStream_t s, int grid_blocks, const DevGraph &g, const DevFrontier &f, uint32_t iter, uint8_t *has, uint8_t *err);
void launch_finalize(hipStream_t s, uint32_t n, const uint8_t *has, const uint8_t *err, uint8_t *perm_out, int32_t *err_out);
void launch_rev_expand(hipStream_t s, int grid_blocks, const DevReverse &r, const DevFrontier &f, uint32_t iter, uint32_t nslots);

}  // namespace acl
