// schema.hpp -- SpiceDB schema subset -> in-memory model used by the plan compiler.
//
// The reference feeds its schema to the embedded SpiceDB as text
// (pkg/spicedb/spicedb.go:19-24, pkg/spicedb/bootstrap.yaml:1-38).  The subset
// accepted here is the one the reference's bootstrap, e2e rules and BASELINE
// configs use: relations with typed subjects (`T`, `T#rel`,
// `with expiration`), permissions built from `+`, `->` / `.any()`, references and
// `nil` -- and, since the reference boots ARBITRARY schemas (spicedb.go:19-24,
// pkg/proxy/options.go:313-316, e2e/embedded_integration_test.go:34-250), intersection
// `&`, exclusion `-` and wildcard subjects `T:*`.  Operator precedence, loosest to
// tightest: `-`, `&`, `+`; same-operator chains associate to the left (EXTERNAL: the
// schema language of github.com/authzed/spicedb, restated, unverified -- DESIGN.md 6).
// Caveats and `.all()` are rejected at load.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace acl {

constexpr int kNoRelation = -1;

struct SubjectClass {  // one allowed subject form of a relation: `stype` or `stype#srel`
    int stype = 0;
    int srel = kNoRelation;  // member index inside stype, or kNoRelation
    bool expiring = false;   // `with expiration`
    bool wildcard = false;   // `stype:*`: one relationship covers every plain subject of the type (srel == kNoRelation)
};

struct Node {  // permission expression tree
    enum Kind { kUnion, kRef, kArrow, kNil, kIntersect, kExclude, kArrowAll } kind = kNil;
    std::vector<Node> kids;   // kUnion, kIntersect: operands; kExclude: {base, subtracted}
    std::string a, b;         // kRef: a ; kArrow: a -> b (a.any(b)) ; kArrowAll: a.all(b), the intersection arrow
    // no `&` / `-` / `.all()` anywhere below: "any HAS wins" (the walk's fast path)
    bool monotone() const {
        if (kind == kIntersect || kind == kExclude || kind == kArrowAll) return false;
        for (const Node &k : kids)
            if (!k.monotone()) return false;
        return true;
    }
};

struct Member {  // a relation or a permission of a definition
    std::string name;
    bool is_permission = false;
    std::vector<SubjectClass> classes;  // relations only
    Node expr;                          // permissions only
    int slot = -1;                      // global (type, member) index
};

struct Definition {
    std::string name;
    std::vector<Member> members;
    std::unordered_map<std::string, int> member_index;
    int find(const std::string &m) const {
        auto it = member_index.find(m);
        return it == member_index.end() ? -1 : it->second;
    }
};

struct Schema {
    std::vector<Definition> defs;
    std::unordered_map<std::string, int> def_index;
    std::vector<int> slot_base;                  // per type: first slot
    std::vector<std::pair<int, int>> slot_owner; // slot -> (type, member)
    int nslots = 0;
    bool has_combine = false;  // some permission uses `&` / `-`

    int type_of(const std::string &n) const {
        auto it = def_index.find(n);
        return it == def_index.end() ? -1 : it->second;
    }
    int slot(int type, int member) const { return slot_base[type] + member; }
    // subject-class id used in requests / probes: (type, rel) -> slot, (type, none) -> nslots + type
    uint32_t subject_key(int stype, int srel) const { return srel == kNoRelation ? (uint32_t)(nslots + stype) : (uint32_t)slot(stype, srel); }
    uint32_t nkeys() const { return (uint32_t)(nslots + defs.size()); }
};

// Parses `text`; on failure returns false and sets `err`.
bool parse_schema(const std::string &text, Schema *out, std::string *err);

}  // namespace acl
