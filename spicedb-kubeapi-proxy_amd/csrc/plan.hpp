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
#include <vector>

#include "store.hpp"

namespace acl {

// ---- device-visible structs (plain, shared with kernels.hip) ----
enum : uint32_t { OP_PROBE = 1u, OP_ENUM = 2u, OP_REFLEX = 4u, OP_PUSH_SAME = 8u };

struct FwdOp {        // 32 B
    uint32_t flags;   // OP_* bits; PROBE|ENUM may be combined (userset class)
    uint32_t dlevel;  // dispatch-depth offset of the state this op belongs to (inlined computed usersets)
    uint32_t off_base;  // index into `off` of this relation's row-offset array
    uint32_t nrows;     // objects covered (ids >= nrows have no relationships)
    uint32_t K;         // subject classes of the relation (row stride)
    uint32_t k;         // subject class of this op
    uint32_t key;       // PROBE/REFLEX: subject key to match; ENUM/PUSH_SAME: target slot
    uint32_t pad;
};
struct SlotProg {       // 16 B
    uint32_t first;     // first op
    uint32_t n_main;    // ops every request runs
    uint32_t n_total;   // n_main + REFLEX ops (only requests whose subject carries a relation)
    uint32_t max_dlevel;  // deepest inlined state
};
struct RevOp {          // 16 B
    uint32_t flags;     // OP_ENUM (reverse row) or OP_PUSH_SAME
    uint32_t roff_base; // index into `roff`
    uint32_t nrows;     // subject id space covered
    uint32_t target;    // slot that becomes true
};
struct RevProg {
    uint32_t first, n;
};

// ---- host-side snapshot ----
struct Snapshot {
    uint64_t revision = 0;     // store revision it was built from
    int64_t valid_lo = 0, valid_hi = 0;  // expiration window of `now`
    // forward
    std::vector<uint32_t> off;    // row offsets (absolute indices into edges)
    std::vector<uint32_t> edges;  // subject ids
    std::vector<FwdOp> ops;
    std::vector<SlotProg> progs;  // [nslots]
    // per type: first slot + member count (request validation on device)
    std::vector<uint32_t> type_slot_base, type_nmembers;
    std::vector<uint32_t> type_nobjects;
    uint32_t nslots = 0, ntypes = 0;
    uint64_t nedges = 0;
    // reverse (built on demand)
    bool has_reverse = false;
    std::vector<uint32_t> roff, redges;
    std::vector<RevOp> rops;
    std::vector<RevProg> rprogs;   // [nslots]: parents of a true state
    std::vector<RevProg> rseeds;   // [nkeys]: seeds for a subject key
    std::vector<uint32_t> slot_bit_base;  // [nslots+1] first bit of each slot's visited bitmap (32-bit aligned)
    std::vector<uint32_t> slot_nobjects;  // [nslots] id space of the slot's type when the reverse rows were built
    uint64_t visited_bits = 0;
};

void build_forward(Store &store, int64_t now, Snapshot *snap);
void build_reverse(Store &store, int64_t now, Snapshot *snap);

}  // namespace acl
