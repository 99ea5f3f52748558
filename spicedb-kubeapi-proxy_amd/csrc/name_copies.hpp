// name_copies.hpp -- keeping a copy of every object type's name-slot array (store.hpp ObjectTable) current from the tables' change lists.  Two
// users: the HBM mirror behind k_resolve_names (engine_names.cpp: hipMalloc / hipMemcpy / a scatter kernel) and its host-side twin behind
// acl_selfcheck_names (engine.cpp: plain vectors), which is how the bookkeeping is tested without a GPU.
#pragma once
#include <cstdint>
#include <vector>

#include "store.hpp"

namespace acl {

struct NameCopyState {  // per type
    size_t cap = 0;        // slots the copy holds
    uint64_t version = 0;  // the table version it reflects
    bool have = false;     // (a type without names has no array; its state is current all the same)
};

// Does bringing the copies up to date REPLACE an array (new size, or a re-hashed table copied over it whole)?  A reader of the copy that
// runs meanwhile could then miss a name that is in the table: the HBM mirror waits for its readers first.  (Names lock held, shared at least.)
inline bool name_copies_need_replacing(const Store &store, const std::vector<NameCopyState> &st) {
    const size_t nt = store.schema().defs.size();
    if (st.size() != nt) return true;
    for (size_t ty = 0; ty < nt; ty++) {
        const ObjectTable &t = store.objects((int)ty);
        if (st[ty].cap != t.slot_count() || (st[ty].version != t.version() && t.changes_are_wholesale())) return true;
    }
    return false;
}

// Ops: int resize(size_t ntypes); int replace(size_t type, size_t slots, const void *bytes) (slots == 0: the type has no names);
//      int patch(size_t type, const std::vector<uint32_t> &slot_indices, const void *all_bytes).  A non-zero return ends the pass (the states of
// the types not yet done stay stale: the next pass picks them up).  *replaced: some array was copied whole (it may have changed place or size).
template <class Ops>
int sync_name_copies(const Store &store, std::vector<NameCopyState> &st, Ops &ops, bool *replaced) {
    const size_t nt = store.schema().defs.size();
    *replaced = false;
    if (st.size() != nt) {
        if (int rc = ops.resize(nt)) return rc;
        st.resize(nt);
        *replaced = true;
    }
    std::vector<uint32_t> idx;
    for (size_t ty = 0; ty < nt; ty++) {
        const ObjectTable &t = store.objects((int)ty);
        NameCopyState &p = st[ty];
        if (p.have && p.version == t.version()) continue;
        bool all = false;
        t.changes(&idx, &all);
        const size_t cap = t.slot_count();
        if (all || cap != p.cap || !p.have) {
            *replaced = true;  // (a whole copy may land in a new array even when the size stays)
            p.have = false;
            p.cap = 0;
            if (int rc = ops.replace(ty, cap, t.slot_bytes())) return rc;
            p.cap = cap;
            p.have = true;
        } else if (!idx.empty()) {
            if (int rc = ops.patch(ty, idx, t.slot_bytes())) return rc;
        }
        p.version = t.version();
    }
    return 0;
}

}  // namespace acl
