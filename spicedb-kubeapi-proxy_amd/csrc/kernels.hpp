// kernels.hpp -- launch interface of the gfx950 frontier-expansion kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.hpp"

namespace acl {

// frontier geometry
constexpr uint32_t kChunk = 1024;         // entries per frontier chunk (16 KiB)
constexpr uint32_t kSegsPerChunk = kChunk / 64;
constexpr uint32_t kMaxFrontierChunks = (1u << 18) - 1;  // 16 B entries of a frontier buffer stay below 4 GiB: the kernels address with 32-bit byte offsets
constexpr uint32_t kMaxLevels = 50;       // dispatch max depth, reference pkg/spicedb/spicedb.go:34
constexpr uint32_t kLevelSlots = 128;     // per-iteration counters (the sharded reverse walk runs two iterations per level)
constexpr uint32_t kMaxShards = 64;      // per-destination export counters (all-to-all exchange)
constexpr uint32_t kStatusWords = 2 * kLevelSlots + 2 + kMaxShards;  // nchunks[] | any[] | overflow | export count | export count per destination
constexpr uint32_t kDeadMeta = 0xFFFFFFFFu;
constexpr int kWavesPerBlock = 4;
constexpr uint32_t kProgLdsEntries = 256;  // ops + progs (32 B each) cached in LDS when they fit

// per-item status byte written by the kernels
enum : uint8_t { ITEM_ERR_NONE = 0, ITEM_ERR_DEPTH = 1, ITEM_ERR_INVALID = 2 };

// frontier entry: one pending sub-check (req, state) -- 16 B, one dwordx4 per lane
//   x = object id, y = request index, z = meta, w = subject id of the request
//   meta = slot[0:13) | level[13:19) | probed[19] | subject key[20:32)
struct DevGraph {
    const uint32_t *meta, *edges, *buckets;  // plan.hpp: row descriptors (uint2), sorted rows, hashed 4-slot buckets (uint4)
    const FwdOp *ops;
    const SlotProg *progs;
    const uint32_t *type_slot_base, *type_nmembers;
    uint32_t nslots, ntypes, nops;
    // combine programs (schemas with `&` / `-`, plan.hpp SlotProg::combine); all unused by the monotone instantiations
    const uint32_t *bexpr = nullptr;   // boolean programs
    uint4 *nodes = nullptr;            // CombineNode records the walk appends (single launch: node_cap per block; level loop: node_cap in all)
    uint32_t *ccount = nullptr;        // level loop: {leaf cells handed out, nodes appended} (the single-launch walk counts in LDS)
    uint32_t node_cap = 0, cell_cap = 0;  // per block (single launch) / in all (level loop)
    uint32_t cell0 = 0;                // index of the first leaf cell in has[] / err[] (= the batch size: cells follow the requests' own)
    uint32_t walk_flags = 0;           // kWalkNoDirect: the single-launch walk builds every task list the general way (set by the host after a walk met a
                                       // pair of segments with more children than the direct form's head-bit window: kernels.hip, process_segment)
};
constexpr uint32_t kWalkNoDirect = 1u;
constexpr uint32_t kOverflowPools = 8u;   // level loop on the SHARDED graph, schemas with `&` / `-`: a shard ran out of combine nodes / leaf cells -- the native loop grows the pools and redoes the batch
constexpr uint32_t kOverflowDirect = 4u;  // single-launch walk's overflow code: "redo, and stop using the direct task lists on this snapshot"
struct DevReverse {
    const uint32_t *rmeta, *redges;  // uint2 {start, end} per (relation, class, subject); resource ids
    const RevOp *rops;
    const RevProg *rprogs, *rseeds;
    const uint64_t *rdest;          // [nslots] other shards holding parent rows of the slot's states (all-to-all form of the sharded walk)
    const uint32_t *slot_bit_base;  // [nslots]
    const uint32_t *slot_nobjects;  // [nslots] id space of the slot's type
    uint32_t *visited;              // [nreq][visited_words]
    uint32_t visited_words;
    uint32_t nslots = 0, nrops = 0;  // sizes of rprogs / rops (the single-launch walk stages them in LDS when they fit)
    // level loop (k_rev_expand) on an unsharded graph: the slots that can lead to the lookup's result slot (Snapshot::rev_useful; bit of slot x < 256) -- ops into any
    // other slot are skipped.  All ones: everything (sharded graphs: a shard's programs are not the whole slot graph).
    uint32_t useful[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
};
constexpr uint32_t kRevLdsRowBytes = 128u << 10;  // the result slot's rows live in LDS up to this size (1 M objects)
constexpr uint32_t kRevLdsSlots = 256, kRevLdsOps = 384;  // what k_rev_local's LDS copy of the reverse programs holds
// Chunk ids: [0, nwaves) are the waves' static first chunks (no allocation), ids >= nwaves come from the
// level's dynamic counter nchunks[iter].  counts[] holds every readable chunk's fill.
struct DevFrontier {
    uint4 *buf[2];
    uint32_t *counts[2];   // per-chunk fill
    uint32_t *nchunks;     // [kLevelSlots] dynamic chunks produced by iteration i (0 = seed)
    uint32_t *any;         // [kLevelSlots] iteration i produced at least one entry
    uint32_t *overflow;    // set when a frontier buffer ran out of chunks
    uint32_t max_chunks;
    uint32_t nwaves;       // waves of every expand launch == number of static chunks
};

// Sharded graph: pending sub-checks whose rows live on another shard leave through `exp`
// (16 B frontier entries, one contiguous buffer per level) and are exchanged by the host between levels.
// `exp_count` keeps counting past `cap` (entries beyond it are dropped) so the host can size a retry.
struct DevShard {
    uint4 *exp = nullptr;
    uint32_t *exp_count = nullptr;   // [1] all-gather form; followed by [kMaxShards] per-destination counters (all-to-all form)
    uint32_t by_dest = 0;            // != 0: `exp` is `world` buffers of `cap` entries, entry -> the buffer of the shard that owns its slot
    uint32_t cap = 0;
    uint32_t rank = 0;
    uint32_t world = 1;
    uint32_t stride = 0;             // by_dest: entries between two destinations' buffers (0 = cap: the buffers are contiguous)
};
// reverse-walk entry flags (meta = slot[0:13) | dist[13:19) | flags)
constexpr uint32_t kRevForeign = 1u << 19;  // state visited on its owner shard: expand it here, do not touch `visited`
constexpr uint32_t kRevVisited = 1u << 20;  // state already passed the visit phase
enum RevPhase : uint32_t { REV_FUSED = 0, REV_VISIT = 1, REV_EXPAND = 2 };

void launch_seed(hipStream_t s, const DevGraph &g, const DevFrontier &f, const uint4 *items, uint32_t n, uint8_t *has, uint8_t *err,
                 const DevShard &sh = DevShard());
void launch_expand(hipStream_t s, const DevGraph &g, const DevFrontier &f, uint32_t iter, uint8_t *has, uint8_t *err,
                   const DevShard &sh = DevShard());
// appends the entries of `in` (an all-gathered export buffer) whose slot this shard owns to the frontier that
// iteration `iter` produced
void launch_import(hipStream_t s, const DevGraph &g, const DevFrontier &f, uint32_t iter, const uint4 *in, uint32_t n, const DevShard &sh);
// native sharded loop: exchange-block headers written on the device (nblocks = 1: all-gather form, exp_count[0]; = world: one per destination,
// exp_count[1 + d]); import of the exchanged blocks + the level's control record {total exported, any produced, overflow code, largest block}.
// hdrs: `world` headers (one per source), data: `world` blocks of `cap` entries; have_data false: the entries were not exchanged this level
void launch_xhdr(hipStream_t s, uint4 *hdr, uint32_t nblocks, const uint32_t *exp_count, const uint32_t *any_iter, const uint32_t *overflow);
void launch_import_gathered(hipStream_t s, const DevGraph &g, const DevFrontier &f, uint32_t iter, const uint4 *hdrs, const uint4 *data, uint32_t world, uint32_t rank,
                            uint32_t cap, bool have_data, uint32_t *ctrl);
void launch_rev_import_gathered(hipStream_t s, const DevReverse &r, const DevFrontier &f, uint32_t iter, const uint4 *hdrs, const uint4 *data, uint32_t world, uint32_t rank,
                                uint32_t cap, bool have_data, uint32_t *ctrl);
void launch_rev_import(hipStream_t s, const DevReverse &r, const DevFrontier &f, uint32_t iter, const uint4 *in, uint32_t n);
void launch_rev_seed(hipStream_t s, const DevFrontier &f, const uint32_t *d_sids, uint32_t n, uint32_t key);
void launch_keep(hipStream_t s, uint32_t k_items, const uint32_t *item_off, const uint8_t *perm, uint8_t *keep_out);
// single-launch Check: units of rpw (<= 256) consecutive requests; block b (of nblocks) walks units b, b + nblocks, ... through every
// level with a private frontier of `cap` entries in each of buf0 / buf1 (regions b * cap); next_unit: zeroed device counter, needed when there are more units than blocks; *overflow != 0 afterwards: redo on the level loop
void launch_check_local(hipStream_t s, const DevGraph &g, const uint4 *items, uint32_t n, uint32_t rpw, uint32_t nblocks, uint32_t *next_unit, uint4 *buf0,
                        uint4 *buf1, uint32_t cap, uint32_t *overflow, uint8_t *has, uint8_t *err, uint8_t *perm_out, int32_t *err_out,
                        uint32_t *max_level = nullptr /* device word (zeroed): atomicMax of the dispatch levels the units needed */,
                        uint32_t nstatic = 0, uint32_t rdyn = 0 /* != 0: only the first nstatic units hold rpw requests, the rest of the batch is cut into units of rdyn
                                                                   handed out through next_unit (which must then be given) */,
                        bool wide = false /* 12 waves per block (and unit) instead of 4: chip-filling batches */,
                        uint32_t skew = 0 /* static units only: unit u holds rpw + skew ... rpw - skew requests, falling with u (items read across PCIe arrive in block order) */,
                        uint32_t *done_ctr = nullptr /* device word, zero between launches */, uint32_t *done_flag = nullptr /* pinned host word (device pointer): the last block to
                        finish stores done_val there, behind a system-scope release of every block's answers -- the host spins on it instead of synchronising the stream */,
                        uint32_t done_val = 0, const uint4 *inline_items_host = nullptr /* HOST pointer to the batch's items: a batch of <= 4 rides in the kernel's arguments
                        and `items` is not read */);
// blocks of the single-launch kernel that are resident at once on this device
int local_grid_blocks(int device, size_t prog_bytes, bool wide = false);  // prog_bytes: (slots + ops) * 32, the kernel's dynamic LDS; wide: the wide (12-wave) instantiation
uint32_t local_unit_max(bool wide = false);  // requests per unit, at most (= threads per block of the single-launch kernel: thread i seeds request i of the unit)
// strikes duplicate (request, state, level) entries of the frontier iteration `iter` produced; table: 2^bits u64 (reset here)
void launch_dedup(hipStream_t s, const DevFrontier &f, uint32_t iter, uint64_t *table, uint32_t bits, bool cells = false);  // cells: the entries carry result cells (combine schemas): `table` is 1.5 x 2^bits words
constexpr uint32_t kDedupBatch = 1u << 14;  // requests per dedup pass (the key holds 14 request bits)
void launch_finalize(hipStream_t s, uint32_t n, const uint8_t *has, const uint8_t *err, uint8_t *perm_out, int32_t *err_out);
// level loop, schemas with `&` / `-`: evaluates the combine nodes of frontier iteration `iter` (call for iter = last .. 1: a node only depends
// on nodes of later iterations); the node count is read from g.ccount[1] on the device
void launch_resolve(hipStream_t s, const DevGraph &g, uint32_t iter, uint8_t *has, uint8_t *err);
// sharded graph, schemas with `&` / `-`: {nodes appended, cells handed out} of this shard as a 16-byte header; the gathered node lists (world blocks of `stride`
// nodes, hdrs[b].x valid ones in block b) resolved for frontier iteration `iter` (members first), on every shard alike
void launch_node_hdr(hipStream_t s, uint4 *hdr, const uint32_t *ccount);
void launch_resolve_gathered(hipStream_t s, const DevGraph &g, const uint4 *nodes, uint32_t stride, const uint4 *hdrs, uint32_t world, uint32_t iter, uint8_t *has, uint8_t *err);
void launch_rev_expand(hipStream_t s, const DevReverse &r, const DevFrontier &f, uint32_t iter, uint32_t phase = REV_FUSED,
                       const DevShard &sh = DevShard());
// single-launch LookupResources: block b walks lookup b (subject sids[b] of class `key`) through every reverse level; visited rows r.visited
// (zeroed by the kernel), private frontier regions of `cap` 8-byte entries per block in buf0 / buf1; the result slot's first `copy_words`
// words go to out_bitmaps + b * out_stride (device or pinned host memory; the rest of the row is zeroed), the id counts to out_counts.
// *status != 0 afterwards (the caller zeroes it; it may live in pinned host memory): redo on the level loop (1) / a row beyond the enumeration limit (2);
// out_counts[b] = ids in the row | reverse levels walked << 56
// Result rows beyond the block's LDS (a type of more than 1 M objects) of a result slot that nothing expands further (the usual case: the permission a list is
// filtered by): with `big` given the ids are marked as BYTES of a per-lookup byte map, the walk defers the heavy rows of the result slot to a chip-wide launch
// and a third launch folds the bytes into the callers' rows from all CUs (kernels.hip RevDefer) -- three launches on `s`, the completion word raised by the last.
struct RevBigRows {
    uint8_t *bytemap = nullptr;     // [n][bytemap_stride], all zero (the third launch zeroes what it found set); bytemap_stride: a multiple of 128 >= the slot's id space
    uint32_t bytemap_stride = 0;
    void *tasks = nullptr;          // [n][task_cap] x 8 bytes
    uint32_t *task_count = nullptr; // [n]
    uint32_t *levels = nullptr;     // [n]
    uint64_t *counts = nullptr;     // [n] device accumulators, zero between launches
    uint32_t task_cap = 0;
    uint32_t defer_min = 0;         // children of one round from which terminal rows are deferred (0 = the default, 4096)
};
constexpr uint32_t kRevUsefulWords = kRevLdsSlots / 32;
struct RevUseful {  // the slots a lookup has to walk (Snapshot::rev_useful's row of its result slot); all ones: everything
    uint32_t w[kRevUsefulWords];
};
// target_slot | kRevTargetSink: nothing above the result slot leads back into it (Snapshot::rev_sink) -- its states end the walk instead of being expanded
constexpr uint32_t kRevTargetSink = 0x80000000u;
void launch_rev_local(hipStream_t s, const DevReverse &r, const uint32_t *sids, uint32_t n, uint32_t key, uint32_t target_slot, void *buf0, void *buf1,
                      uint32_t cap, uint32_t *out_bitmaps, uint32_t out_stride, uint32_t copy_words, uint64_t *out_counts, uint32_t *status,
                      uint32_t lds_row_words /* words covering the result slot's id space: kept in LDS when <= kRevLdsRowBytes, else (or 0) in r.visited */,
                      uint32_t *done_ctr = nullptr, uint32_t *done_flag = nullptr, uint32_t done_val = 0 /* as launch_check_local: the last block stores done_val into the pinned
                      word done_flag behind a system-scope release of every block's rows, counts and status */,
                      const RevBigRows *big = nullptr, const RevUseful *useful = nullptr /* NULL: every slot */);
// blocks per expand launch for this device (all co-resident); nwaves = blocks * kWavesPerBlock
int expand_grid_blocks(int device);

}  // namespace acl
