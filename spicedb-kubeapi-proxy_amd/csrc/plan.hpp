// plan.hpp -- schema + relationship store -> (HBM snapshot, frontier programs).
//
// This is the "Zanzibar relationship graph held in HBM as a CSR" of the
// north star.  What SpiceDB does per dispatched sub-check at run time (look up
// the relation's rewrite, iterate the tuples of one object#relation; the engine
// reached from pkg/authz/check.go:48 and lookups.go:65) is split here into
//   * data:     per-relation CSR rows `object x subject-class -> sorted subject ids`
//   * programs: per (type, relation|permission) a flattened list of row operations
//               (probe for the terminal subject / enumerate usersets / follow an arrow),
//               computed-userset rewrites inlined with their dispatch-depth offset.
// The kernels (kernels.hip) interpret the programs against the data.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "store.hpp"

namespace acl {

// ---- device-visible structs (plain, shared with kernels.hip) ----
enum : uint32_t {
    OP_PROBE = 1u,      // membership test of the request's subject in a SORTED sub-row (binary search)
    OP_ENUM = 2u,       // every subject of a SORTED sub-row becomes a child state
    OP_REFLEX = 4u,     // subject == this very object#relation
    OP_PUSH_SAME = 8u,  // non-inlined computed userset: child state on the same object
    OP_PROBE_HASH = 16u, // membership test in a membership-only class: SUBJECT-indexed hashed rows (4-slot buckets)
    OP_LEAFBIT = 32u,    // with OP_ENUM: bit 31 of every edge says "this child has nothing to enumerate"
    OP_NOMARK = 128u,    // reverse seed ops (RevOp): the children of this op are the ONLY states of their slot the walk can ever produce, each once
                         // (no other op targets the slot; ids of a reverse row are distinct): no visited bit is needed to tell a first visit
    OP_ALL = 256u,       // with OP_ENUM, inside the leaf of an intersection arrow `a.all(b)`: every child answers a result cell of its OWN, folded
                         // into the leaf's cell by a member node (kernels.hip kAllBit; BX_LEAF_ALL)
    OP_DEAD = 512u,      // reverse ops, set in k_rev_local's LDS copy only: the op's target slot cannot lead to the lookup's result slot (Snapshot::rev_useful) -- skipped
    OP_WILD = 64u        // with OP_PROBE_HASH: the row probed is the one of the class's wildcard subject `T:*` (its id in FwdOp::K), whoever the
                         // request's subject is; reverse ops (RevOp): the row read is the wildcard subject's (roff_base points at it), whatever the seed's id
};
// Hashed rows: 4-slot buckets (16 B) of ids.  Round 5: a row is SEEDED SINGLE-CHOICE wherever that can be had -- the builder searches the
// row's 8-bit seed for a placement in which no bucket overflows, so a membership test is ONE 16-byte gather (an id lives in bucket
// hrow_bucket(id) or nowhere).  Rows of a few dozen ids (a user's groups, a user's pods) find a seed at the old load of 0.75; longer rows get
// more buckets (load down to 1/3) and rows for which that is not enough keep the two-choice (cuckoo) placement of rounds 1-4: bucket h1 or h2,
// two independent gathers, which the kernels issue only for wave steps that hold such a row.  Either way there is never a probing chain whose
// longest lane stalls the wave.
//   descriptor {x, y}: x = first bucket, y = buckets[0:23) | two-choice[23] | seed[24:32); y == 0: the subject has no row
#if defined(__HIPCC__)
#define ACL_HD __host__ __device__
#else
#define ACL_HD
#endif
constexpr uint32_t kRowNbMask = 0x7FFFFFu;   // buckets of one row (< 2^23: 25 M ids of one subject; the builder fails loudly beyond)
constexpr uint32_t kRowTwoBit = 1u << 23;    // two-choice row (seed 0)
ACL_HD inline uint32_t hrow_nb(uint32_t y) { return y & kRowNbMask; }
ACL_HD inline uint32_t hrow_pack(uint32_t nb, uint32_t seed, bool two) { return nb | (two ? kRowTwoBit : 0u) | (seed << 24); }
#ifndef ACL_ROW_HASH
#define ACL_ROW_HASH 1  // 1 (round 6): FAST rows -- single-choice, fewer than 2^16 buckets: every row of the BASELINE graphs -- hash in full-rate 24-bit multiplies;
                        // 0 = the 32-bit hash of round 5 for every row (A/B builds; the host builds the rows with whatever this says)
#endif
// Round 6: the bucket of `id` is computed once per CHILD in the walk's hottest loop, and round 5's hash cost it three quarter-rate instructions
// (v_mul_lo_u32 x 2, v_mul_hi_u32: 48 issue cycles of a wave64 SIMD) -- profiles/r05_pmc_c4.md: the walk is instruction-issue bound.  A FAST row
// (y & kRowSlowMask == 0: single-choice and nb < 2^16) uses
//     t = id ^ (id >> 12) ^ seed        the high id bits folded onto the low 24, the row's seed xor-ed into the low byte
//     h = lo24(t) * 0x9E3779            v_mul_u32_u24: full rate; bits 8..23 of the product = the classic multiplicative hash of a 24-bit word
//     bucket = (bits(h, 8, 16) * nb) >> 16   v_mul_u32_u24 again: a 16-bit fraction times nb < 2^16 stays inside 32 bits
// -- eight full-rate instructions with the address arithmetic.  hrow_fast() is IN RANGE for every row (nb is cut to 16 bits), so the kernels issue it
// for all lanes unconditionally and redo the rare slow rows (two-choice, or 2^16 buckets and more: > 196 000 ids of one subject) behind a ballot.
constexpr uint32_t kRowSlowMask = ACL_ROW_HASH ? 0x00FF0000u : 0xFFFFFFFFu;
ACL_HD inline bool hrow_is_slow(uint32_t y) { return (y & kRowSlowMask) != 0u; }
ACL_HD inline uint32_t hrow_fast(uint32_t id, uint32_t y) {
    const uint32_t t = id ^ (id >> 12) ^ (y >> 24);
#if defined(__HIP_DEVICE_COMPILE__)
    // (said in the ISA's own words: only bits 8..23 of the product are used and those depend on the low 24 bits of t alone, so the compiler drops every
    //  mask put on t -- __umul24's included -- and is then left with a 32 x 32-bit multiply, quarter rate)
    uint32_t h, b;
    asm("v_mul_u32_u24 %0, 0x9e3779, %1" : "=v"(h) : "v"(t));
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(b) : "v"((h >> 8) & 0xFFFFu), "v"(y & 0xFFFFu));
    return b >> 16;
#else
    const uint32_t h = (t & 0xFFFFFFu) * 0x9E3779u;  // (24 x 24 bits: the low 32 bits of the product, what v_mul_u32_u24 returns)
    return (((h >> 8) & 0xFFFFu) * (y & 0xFFFFu)) >> 16;
#endif
}
ACL_HD inline uint32_t hrow_slow(uint32_t id, uint32_t y) {
    return (uint32_t)(((uint64_t)((id ^ ((y >> 24) * 0x85EBCA6Bu)) * 0x9E3779B1u) * (y & kRowNbMask)) >> 32);
}
// the bucket of `id` in a single-choice row -- and the first choice in a two-choice row (seed 0)
ACL_HD inline uint32_t hrow_bucket(uint32_t id, uint32_t y) { return hrow_is_slow(y) ? hrow_slow(id, y) : hrow_fast(id, y); }
// the second choice of a two-choice row
ACL_HD inline uint32_t hrow_bucket2(uint32_t id, uint32_t y, uint32_t h1) {
    const uint32_t nb = y & kRowNbMask;
    const uint32_t b = (uint32_t)(((uint64_t)((id ^ 0x5bd1e995u) * 0x85EBCA6Bu) * nb) >> 32);
    return b != h1 ? b : (h1 + 1 == nb ? 0u : h1 + 1);  // nb == 1: both are bucket 0
}
constexpr uint32_t kLeafBit = 0x80000000u;  // object ids are < 2^31
constexpr uint32_t kIdMask = 0x7FFFFFFFu;

struct FwdOp {        // 32 B
    uint32_t flags;   // OP_* bits; PROBE|ENUM may be combined (userset class)
    uint32_t dlevel;  // dispatch-depth offset of the state this op belongs to (inlined computed usersets)
    uint32_t base;    // index (uint2 units) into `meta`: sorted ops -> the relation's per-object row descriptors,
                      // PROBE_HASH -> the class's per-SUBJECT row descriptors
    uint32_t nrows;   // ids covered by those descriptors (ids >= nrows have no relationships)
    uint32_t K;       // sorted subject classes of the relation (row stride); PROBE_HASH | WILD: the wildcard subject's id
    uint32_t k;       // sorted-class index of this op
    uint32_t key;     // PROBE*/REFLEX: subject key to match; ENUM/PUSH_SAME: target slot
    uint32_t leaf;    // 0: a hit answers the entry's own result cell; L > 0 (programs with `&` / `-` only): cell L - 1 of the state's leaf cells
};
struct SlotProg {       // 32 B.  Op order: [probe-only ops][ops that may create children][REFLEX ops]
    uint32_t first;     // first op
    uint32_t n_probe;   // leading ops that only probe (skipped for a state whose probes were done by its parent)
    uint32_t n_main;    // n_probe + child-creating ops
    uint32_t n_total;   // n_main + REFLEX ops (only requests whose subject carries a relation)
    uint32_t max_dlevel;  // deepest inlined state
    uint32_t owner;       // shard that holds this slot's type (rows + program); 0 when the graph is not sharded
    // Rewrites with intersection / exclusion ("combine" programs; reference pkg/spicedb/spicedb.go:19-24 boots any schema): the state's value is
    // a boolean program over LEAVES -- maximal union-only sub-expressions, each answered by an ordinary monotone walk into a result cell of
    // its own -- evaluated when the whole walk is over (kernels.hip, resolve).  combine = index of that program in Snapshot::bexpr, 0 = none.
    uint32_t combine;
    uint32_t nleaves;
};
// Snapshot::bexpr at SlotProg::combine: [ntokens][deepest inlined dispatch offset of leaf 0 (the direct ops) .. leaf nleaves][tokens...], postfix:
constexpr uint32_t BX_LEAF = 1u << 24;  // | leaf number (1-based): push the leaf cell's value
constexpr uint32_t BX_OR = 2u << 24;    // | n: HAS > ERR > NO over the n topmost values
constexpr uint32_t BX_AND = 3u << 24;   // | n: NO > ERR > HAS
constexpr uint32_t BX_EXCL = 4u << 24;  // base, subtracted: base unless HAS; then subtracted ERR -> ERR, HAS -> NO, NO -> HAS
constexpr uint32_t BX_LEAF_ALL = 5u << 24;  // | leaf number: the cell of an intersection arrow `a.all(b)` -- has = some child said HAS, err bit 7 = some child
                                            // said NO, err low bits = some child erred: NO if bit 7 or no child, else ERR, else HAS (kAllNoBit)
constexpr uint8_t kAllNoBit = 0x80u;
constexpr uint32_t kMaxLeaves = 30;     // per state (the resolve keeps its value stack in one 64-bit register, two bits per value)
struct CombineNode {  // 16 B: one visited state with a combine program (written by the walk, read by the resolve)
    uint32_t out;     // result cell the state's value is OR-ed into
    uint32_t cells;   // first of its nleaves leaf cells
    uint32_t slot_iter;  // slot | frontier iteration << 16: a node only depends on nodes of LATER iterations
    uint32_t pad;
};
struct RevOp {          // 16 B
    uint32_t flags;     // OP_ENUM (reverse row) or OP_PUSH_SAME
    uint32_t roff_base; // first row descriptor (uint2 {start, end} per subject id) in `rmeta`
    uint32_t nrows;     // subject id space covered
    uint32_t target;    // slot that becomes true
};
struct RevProg {
    uint32_t first, n;  // bit 31 of n (kRevRemoteBit): other shards hold parent rows of this state too
};
constexpr uint32_t kRevRemoteBit = 0x80000000u;

// ---- host-side layout of one relation's rows (kept with the snapshot so that writes can patch it in place)
struct ClassLayout {
    bool live = false;    // >= 1 live relationship when the snapshot was built (decides whether programs hold an op for it)
    bool hashed = false;  // membership-only class: subject-indexed hashed rows
    uint32_t ks = 0;      // sorted-class index inside the relation's row descriptors
    uint32_t smeta_base = 0, nsubjects = 0;  // hashed: per-subject descriptors (uint2 units) and their count (with headroom)
};
struct RelLayout {
    uint32_t meta_base = 0, nrows = 0, Ks = 0;  // meta_base in uint2 units; nrows includes headroom for new objects
    std::vector<ClassLayout> cls;
};
struct Patch {  // a region of a snapshot array that changed on the host and must be re-uploaded
    enum Array { META = 0, EDGES = 1, BUCKETS = 2, OPS = 3, RMETA = 4, REDGES = 5 } array;
    size_t off, n;  // in elements of that array (u32 words; FwdOp for OPS)
};

// ---- host-side snapshot ----
struct Snapshot {
    uint64_t revision = 0;     // store revision it was built from
    int64_t valid_lo = 0, valid_hi = 0;  // expiration window of `now`
    int64_t patch_lo = 0, patch_hi = 0;  // ... as it was before the last patch_forward (patch_reverse replays the same expiry crossings)
    // forward
    std::vector<uint32_t> meta;     // uint2 {start, end}: per (object, sorted class) into edges; per (hashed class, SUBJECT) into buckets
    std::vector<uint32_t> edges;    // SORTED sub-rows: subject ids ascending (| kLeafBit, see OP_LEAFBIT)
    std::vector<uint32_t> buckets;  // hashed rows: uint4 buckets of RESOURCE ids of one subject, empty slot = 0xFFFFFFFF
    std::vector<FwdOp> ops;
    std::vector<SlotProg> progs;  // [nslots]
    std::vector<uint32_t> bexpr;  // boolean programs of the combine slots (word 0 unused: SlotProg::combine == 0 means none)
    bool has_combine = false;     // some slot's rewrite uses `&` / `-`: evaluations run the kernels' combine instantiations
    std::vector<uint8_t> slot_nonmono;  // [nslots] the slot's value can depend on a combine program: LookupResources = candidates + a forward Check
    // [nslots] 1: a Check of this slot could run into the dispatch-depth limit -- its dependencies (userset subjects, references, arrows) reach a cycle
    // (`group#member@group#member`) or a chain of more than 25 of them.  0: whatever the relationships are, no Check of it ends in a depth error, so
    // "the reverse walk's bit is not set" and "NO_PERMISSION without an error" are the same statement (engine.cpp: CheckBulkPermissions by one reverse walk).
    std::vector<uint8_t> slot_deep;
    // per type: first slot + member count (request validation on device)
    std::vector<uint32_t> type_slot_base, type_nmembers;
    std::vector<uint32_t> type_nobjects;
    std::vector<RelLayout> lay;  // [nslots]
    uint64_t garbage_words = 0;  // edge / bucket words orphaned by row relocations since the build
    // Rows that a patch had to move get room to grow (1.5x): start offset -> capacity in words.  Host-only: the kernels read
    // [start, end) of a descriptor and never see the slack behind it.  Without it a namespace that gains pods one write at a
    // time (or a group that gains members) would be copied whole, and orphan its old copy, on every single write.
    std::unordered_map<uint32_t, uint32_t> edge_cap, redge_cap;
    uint64_t patched = 0;        // relationships patched in since the build
    uint32_t nslots = 0, ntypes = 0;
    uint64_t nedges = 0;        // relationships in the store
    uint64_t nedges_local = 0;  // ... whose resource type this shard owns
    std::vector<uint32_t> type_owner;  // [ntypes] shard of each type
    // reverse (built on demand)
    bool has_reverse = false;
    std::vector<uint32_t> rmeta, redges;  // reverse rows: uint2 {start, end} per (relation, class, SUBJECT id) into redges (resource ids)
    struct RevLayout { bool any = false; uint32_t base = 0, nrows = 0; };  // base in uint2 units; nrows includes headroom
    std::vector<std::vector<RevLayout>> rlay;  // [slot][class]
    std::vector<RevOp> rops;
    std::vector<RevProg> rprogs;   // [nslots]: parents of a true state
    // [nslots][kRevUsefulWords]: bit x of row t -- slot x is t itself or some chain of parent programs leads from x to t: what a lookup of t has to walk.  States of
    // any other slot can never make a state of t true; their ops are dead for that lookup (a lookup of pod#creator does not walk the user's groups, a lookup of
    // group#member not the namespaces and pods the groups are named on).
    std::vector<uint32_t> rev_useful;
    std::vector<uint8_t> rev_sink;  // [nslots]: no chain of parent programs leads from the slot back into it -- as a lookup's RESULT slot its states need no expansion
    std::vector<RevProg> rseeds;   // [nkeys]: seeds for a subject key
    std::vector<uint64_t> rdest;   // [nslots]: the OTHER shards that hold parent rows of a state of this slot (bit per shard < 64; sharded graphs)
    std::vector<uint32_t> slot_bit_base;  // [nslots+1] first bit of each slot's visited bitmap (32-bit aligned)
    std::vector<uint32_t> slot_nobjects;  // [nslots] id space of the slot's type the visited bitmaps cover (with headroom for new objects)
    uint64_t visited_bits = 0;
};

// Graph partition of the north star's multi-GPU configuration: rows (and programs) of an object type live on
// shard mix(fnv1a(type name)) mod world (plan.cpp shard_of_type).  world == 1: everything is local.
struct ShardSpec {
    uint32_t rank = 0, world = 1;
};
uint32_t shard_of_type(const std::string &type_name, uint32_t world);

void build_forward(Store &store, int64_t now, Snapshot *snap, ShardSpec shard = ShardSpec());
// Brings a snapshot built at an older store revision up to date IN PLACE from the store's change feed: hashed rows
// are re-hashed per subject, sorted rows are shrunk in place or relocated to the end of `edges`, leaf flags that
// may have gone stale are switched off per program op.  Returns false when the change cannot be expressed as a
// patch (bulk load, a class becoming live, table headroom exhausted, too many changes): rebuild instead.
// On success `patches` lists the regions to re-upload and the reverse rows (if any) are invalidated.
// max_changes: how many feed entries a patch may carry (0 = the default bound beyond which a rebuild is cheaper; the adoption of
// a background-built snapshot passes a larger one: there the alternative is not a rebuild but throwing a finished build away).
// A `now` outside the snapshot's expiration window is part of the feed: the relationships whose expiry lies between the window and
// `now` are patched out (or, for a clock set back, in) like deletions -- an idempotency key running out (bootstrap.yaml:34-36) is not
// a reason to rebuild 10 M relationships.
bool patch_forward(Store &store, int64_t now, Snapshot *snap, ShardSpec shard, std::vector<Patch> *patches, size_t max_changes = 0);
// the relationships whose liveness differs between a snapshot valid for [lo, hi) and `now` (appended to *ch; nothing when now is inside)
void expiry_crossings(const Store &store, int64_t lo, int64_t hi, int64_t now, std::vector<Store::Change> *ch);
// Same for the reverse rows (LookupResources); call after a successful patch_forward with the same feed position
// (`from_revision` = the snapshot's revision BEFORE patch_forward).  false: rebuild the reverse rows instead.
bool patch_reverse(Store &store, int64_t now, uint64_t from_revision, Snapshot *snap, ShardSpec shard, std::vector<Patch> *patches);
uint32_t with_headroom(uint32_t n);
bool verify_snapshot(Store &store, int64_t now, const Snapshot &snap, ShardSpec shard, std::string *why);
void build_reverse(Store &store, int64_t now, Snapshot *snap, ShardSpec shard = ShardSpec());

}  // namespace acl
