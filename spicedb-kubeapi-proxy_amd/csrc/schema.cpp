// schema.cpp -- recursive-descent parser for the supported SpiceDB schema subset.
// Reference input it must accept verbatim: pkg/spicedb/bootstrap.yaml:2-38.
#include "schema.hpp"

#include <cctype>
#include <stdexcept>

namespace acl {
namespace {

struct Tok {
    enum T { kEnd, kIdent, kPunct, kArrow } t = kEnd;
    std::string s;
    size_t pos = 0;
};

class Lexer {
  public:
    explicit Lexer(const std::string &src) : src_(src) { advance(); }
    const Tok &peek() const { return cur_; }
    Tok take() {
        Tok t = cur_;
        advance();
        return t;
    }
    bool is_punct(char c) const { return cur_.t == Tok::kPunct && cur_.s[0] == c; }
    bool is_word(const char *w) const { return cur_.t == Tok::kIdent && cur_.s == w; }
    [[noreturn]] void fail(const std::string &msg) const {
        size_t line = 1;
        for (size_t i = 0; i < cur_.pos && i < src_.size(); i++) line += src_[i] == '\n';
        throw std::runtime_error("schema line " + std::to_string(line) + ": " + msg + (cur_.t == Tok::kEnd ? " (at end of input)" : " (at `" + cur_.s + "`)"));
    }
    void expect_punct(char c) {
        if (!is_punct(c)) fail(std::string("expected `") + c + "`");
        advance();
    }
    std::string expect_ident(const char *what) {
        if (cur_.t != Tok::kIdent) fail(std::string("expected ") + what);
        return take().s;
    }

  private:
    void skip_space() {
        for (;;) {
            while (i_ < src_.size() && std::isspace((unsigned char)src_[i_])) i_++;
            if (i_ + 1 < src_.size() && src_[i_] == '/' && src_[i_ + 1] == '/') {
                while (i_ < src_.size() && src_[i_] != '\n') i_++;
            } else if (i_ + 1 < src_.size() && src_[i_] == '/' && src_[i_ + 1] == '*') {
                size_t e = src_.find("*/", i_ + 2);
                i_ = e == std::string::npos ? src_.size() : e + 2;
            } else {
                return;
            }
        }
    }
    void advance() {
        skip_space();
        cur_ = Tok();
        cur_.pos = i_;
        if (i_ >= src_.size()) return;
        char c = src_[i_];
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t b = i_;
            // definition names may carry a prefix: tenant/name
            while (i_ < src_.size() && (std::isalnum((unsigned char)src_[i_]) || src_[i_] == '_' || src_[i_] == '/')) i_++;
            cur_.t = Tok::kIdent;
            cur_.s = src_.substr(b, i_ - b);
        } else if (c == '-' && i_ + 1 < src_.size() && src_[i_ + 1] == '>') {
            cur_.t = Tok::kArrow;
            cur_.s = "->";
            i_ += 2;
        } else {
            cur_.t = Tok::kPunct;
            cur_.s = std::string(1, c);
            i_++;
        }
    }
    const std::string &src_;
    size_t i_ = 0;
    Tok cur_;
};

Node parse_expr(Lexer &lx);

Node parse_operand(Lexer &lx) {
    if (lx.is_punct('(')) {
        lx.take();
        Node n = parse_expr(lx);
        lx.expect_punct(')');
        return n;
    }
    std::string first = lx.expect_ident("a relation or permission name");
    Node n;
    if (first == "nil") {
        n.kind = Node::kNil;
        return n;
    }
    if (lx.peek().t == Tok::kArrow) {
        lx.take();
        n.kind = Node::kArrow;
        n.a = first;
        n.b = lx.expect_ident("a name after `->`");
        return n;
    }
    if (lx.is_punct('.')) {  // a.any(b) is a->b; a.all(b) is an intersection arrow
        lx.take();
        std::string fn = lx.expect_ident("`any` or `all`");
        if (fn != "any" && fn != "all") lx.fail("unsupported: `." + fn + "()` arrows");
        lx.expect_punct('(');
        n.kind = fn == "all" ? Node::kArrowAll : Node::kArrow;
        n.a = first;
        n.b = lx.expect_ident("a name inside `.any()` / `.all()`");
        lx.expect_punct(')');
        return n;
    }
    n.kind = Node::kRef;
    n.a = first;
    return n;
}

// one binary level of the expression grammar: operands from `sub`, joined by `op`, left-associative.  Unions and intersections flatten
// (a + b + c is one node of three operands); an exclusion stays binary {base, subtracted}.
template <typename Sub>
Node parse_level(Lexer &lx, char op, Node::Kind kind, Sub sub) {
    Node lhs = sub(lx);
    while (lx.is_punct(op)) {
        lx.take();
        Node rhs = sub(lx);
        if (lhs.kind == kind && kind != Node::kExclude) {
            lhs.kids.push_back(std::move(rhs));
        } else {
            Node u;
            u.kind = kind;
            u.kids.push_back(std::move(lhs));
            u.kids.push_back(std::move(rhs));
            lhs = std::move(u);
        }
    }
    return lhs;
}
// precedence, loosest first: exclusion, intersection, union (schema.hpp)
Node parse_union(Lexer &lx) { return parse_level(lx, '+', Node::kUnion, parse_operand); }
Node parse_inter(Lexer &lx) { return parse_level(lx, '&', Node::kIntersect, parse_union); }
Node parse_expr(Lexer &lx) { return parse_level(lx, '-', Node::kExclude, parse_inter); }

struct PendingClass {
    std::string type, rel;
    bool expiring;
    bool wildcard;
};

void check_refs(const Schema &s, const Definition &d, const Member &m, const Node &n) {
    switch (n.kind) {
        case Node::kUnion:
        case Node::kIntersect:
        case Node::kExclude:
            for (const Node &k : n.kids) check_refs(s, d, m, k);
            break;
        case Node::kRef:
            if (d.find(n.a) < 0) throw std::runtime_error("schema: permission `" + d.name + "#" + m.name + "` references unknown `" + n.a + "`");
            break;
        case Node::kArrow:
        case Node::kArrowAll: {
            int ts = d.find(n.a);
            if (ts < 0 || d.members[ts].is_permission)
                throw std::runtime_error("schema: permission `" + d.name + "#" + m.name + "` has an arrow over `" + n.a + "`, which is not a relation");
            for (const SubjectClass &c : d.members[ts].classes)
                if (c.wildcard)
                    throw std::runtime_error("schema: permission `" + d.name + "#" + m.name + "` has an arrow over `" + n.a + "`, which allows wildcard subjects");
            // a.all(b) fails CLOSED: a plain arrow skips tupleset subjects whose type has no `b`; an intersection arrow that skipped them
            // could grant what the real engine denies (its rule there is unverified), so such a schema is refused at load
            if (n.kind == Node::kArrowAll)
                for (const SubjectClass &c : d.members[ts].classes)
                    if (s.defs[c.stype].find(n.b) < 0)
                        throw std::runtime_error("schema: permission `" + d.name + "#" + m.name + "`: `" + n.a + ".all(" + n.b + ")` over subject type `" + s.defs[c.stype].name +
                                                 "`, which has no `" + n.b + "`");
            break;
        }
        case Node::kNil: break;
    }
}

// result cells a permission's boolean program needs (plan.cpp cuts everything under `&` / `-` into union-only leaves)
size_t combine_leaves(const Node &n) {
    if (n.monotone()) return 1;
    size_t c = 0;
    bool mono = false;
    for (const Node &k : n.kids) {
        if (n.kind == Node::kUnion && k.monotone()) mono = true;
        else c += combine_leaves(k);
    }
    return c + (mono ? 1 : 0);
}

}  // namespace

bool parse_schema(const std::string &text, Schema *out, std::string *err) {
    try {
        Schema s;
        std::vector<std::vector<std::vector<PendingClass>>> pending;  // [type][member] -> classes
        Lexer lx(text);
        while (lx.peek().t != Tok::kEnd) {
            if (lx.is_word("use")) {  // `use expiration`
                lx.take();
                lx.expect_ident("a feature name after `use`");
                continue;
            }
            if (lx.is_word("caveat")) lx.fail("unsupported: caveats");
            if (!lx.is_word("definition")) lx.fail("expected `definition`");
            lx.take();
            Definition d;
            d.name = lx.expect_ident("a definition name");
            if (s.def_index.count(d.name)) lx.fail("duplicate definition `" + d.name + "`");
            std::vector<std::vector<PendingClass>> dpend;
            lx.expect_punct('{');
            while (!lx.is_punct('}')) {
                bool is_perm = lx.is_word("permission");
                if (!is_perm && !lx.is_word("relation")) lx.fail("expected `relation`, `permission` or `}`");
                lx.take();
                Member m;
                m.is_permission = is_perm;
                m.name = lx.expect_ident("a name");
                if (d.member_index.count(m.name)) lx.fail("duplicate member `" + m.name + "`");
                std::vector<PendingClass> classes;
                if (is_perm) {
                    lx.expect_punct('=');
                    m.expr = parse_expr(lx);
                } else {
                    lx.expect_punct(':');
                    for (;;) {
                        PendingClass pc;
                        pc.type = lx.expect_ident("a subject type");
                        pc.expiring = false;
                        pc.wildcard = false;
                        if (lx.is_punct(':')) {  // `T:*`
                            lx.take();
                            lx.expect_punct('*');
                            pc.wildcard = true;
                        } else if (lx.is_punct('#')) {
                            lx.take();
                            pc.rel = lx.expect_ident("a subject relation after `#`");
                        }
                        if (lx.is_word("with")) {
                            lx.take();
                            std::string trait = lx.expect_ident("`expiration`");
                            if (trait != "expiration") lx.fail("unsupported: caveated relation (`with " + trait + "`)");
                            if (lx.is_word("and")) lx.fail("unsupported: caveated relation");
                            pc.expiring = true;
                        }
                        classes.push_back(pc);
                        if (!lx.is_punct('|')) break;
                        lx.take();
                    }
                }
                d.member_index[m.name] = (int)d.members.size();
                d.members.push_back(std::move(m));
                dpend.push_back(std::move(classes));
            }
            lx.expect_punct('}');
            s.def_index[d.name] = (int)s.defs.size();
            s.defs.push_back(std::move(d));
            pending.push_back(std::move(dpend));
        }
        // slots
        for (size_t t = 0; t < s.defs.size(); t++) {
            s.slot_base.push_back(s.nslots);
            for (size_t m = 0; m < s.defs[t].members.size(); m++) {
                s.defs[t].members[m].slot = s.nslots++;
                s.slot_owner.emplace_back((int)t, (int)m);
            }
        }
        if (s.nslots + s.defs.size() >= 4095) throw std::runtime_error("schema: too many relations/permissions (limit 4094 incl. types)");
        // resolve subject classes
        for (size_t t = 0; t < s.defs.size(); t++)
            for (size_t m = 0; m < s.defs[t].members.size(); m++) {
                Member &mem = s.defs[t].members[m];
                for (const PendingClass &pc : pending[t][m]) {
                    SubjectClass sc;
                    sc.stype = s.type_of(pc.type);
                    if (sc.stype < 0) throw std::runtime_error("schema: relation `" + s.defs[t].name + "#" + mem.name + "` allows unknown type `" + pc.type + "`");
                    if (!pc.rel.empty()) {
                        sc.srel = s.defs[sc.stype].find(pc.rel);
                        if (sc.srel < 0) throw std::runtime_error("schema: relation `" + s.defs[t].name + "#" + mem.name + "` allows unknown `" + pc.type + "#" + pc.rel + "`");
                    }
                    sc.expiring = pc.expiring;
                    sc.wildcard = pc.wildcard;
                    bool dup = false;
                    for (SubjectClass &e : mem.classes)
                        if (e.stype == sc.stype && e.srel == sc.srel && e.wildcard == sc.wildcard) { e.expiring |= sc.expiring; dup = true; }
                    if (!dup) mem.classes.push_back(sc);
                }
            }
        for (const Definition &d : s.defs)
            for (const Member &m : d.members)
                if (m.is_permission) {
                    check_refs(s, d, m, m.expr);
                    if (!m.expr.monotone()) s.has_combine = true;
                    if (!m.expr.monotone() && combine_leaves(m.expr) > 30)  // (plan.hpp kMaxLeaves)
                        throw std::runtime_error("schema: permission `" + d.name + "#" + m.name + "` combines more than 30 operands under `&` / `-`");
                }
        *out = std::move(s);
        return true;
    } catch (const std::exception &e) {
        if (err) *err = e.what();
        return false;
    }
}

}  // namespace acl
