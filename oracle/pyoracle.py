"""pyoracle -- second, independent CPU ORACLE (test infrastructure, NOT product code).

A deliberately tiny pure-Python restatement of the same SpiceDB semantics as
oracle/acl_oracle.c (see that file's header for the reference call sites:
pkg/authz/check.go:48, pkg/authz/lookups.go:65, pkg/authz/watch.go:50, engine
config pkg/spicedb/spicedb.go:34,60).  It exists because the real engine
(github.com/authzed/spicedb, go.mod:9) cannot run here, so two independently
written restatements arbitrate each other under hypothesis-generated graphs.

Written in a different style on purpose: regex schema parser, dict-of-sets
store, recursive evaluation returning a 3-valued result.
Pure-Python loops: small cases only.

Round 4: intersection `&`, exclusion `-` (precedence loosest to tightest `-`, `&`, `+`;
three-valued results: union HAS > ERR > NO, intersection NO > ERR > HAS, exclusion
base-first) and wildcard subjects `T:*` -- the same EXTERNAL, unverified restatement
of SpiceDB's `all` / `difference` as the C oracle's header describes, written here as
a precedence-climbing parser and table lookups instead of nested ifs.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field

MAX_DEPTH = 50  # pkg/spicedb/spicedb.go:34

NO, HAS, ERR = "NO", "HAS", "ERR"


class SchemaError(ValueError):
    pass


@dataclass
class Relation:
    name: str
    allowed: list  # [(subject_type, subject_relation_or_None, expiring)]


@dataclass
class Permission:
    name: str
    expr: tuple  # ('union' | 'inter' | 'excl', a, b) | ('ref', name) | ('arrow' | 'arrow_all', tupleset, computed) | ('nil',)


@dataclass
class Definition:
    name: str
    members: dict = field(default_factory=dict)  # name -> Relation | Permission


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _parse_expr(tokens: list, pos: int):
    def term(p):
        t = tokens[p]
        if t == "(":
            e, p = _parse_expr(tokens, p + 1)
            if tokens[p] != ")":
                raise SchemaError("expected )")
            return e, p + 1
        if not re.fullmatch(r"[A-Za-z_][\w/]*", t):
            raise SchemaError(f"unexpected token {t!r}")
        if t == "nil":
            return ("nil",), p + 1
        if p + 1 < len(tokens) and tokens[p + 1] == "->":
            return ("arrow", t, tokens[p + 2]), p + 3
        if p + 1 < len(tokens) and tokens[p + 1] == ".":
            if tokens[p + 2] not in ("any", "all"):
                raise SchemaError("unsupported arrow function (any / all)")
            return ("arrow" if tokens[p + 2] == "any" else "arrow_all", t, tokens[p + 4]), p + 6
        return ("ref", t), p + 1

    # precedence climbing: a level's operands are the next-tighter level's expressions
    levels = [("-", "excl"), ("&", "inter"), ("+", "union")]

    def level(i, p):
        if i == len(levels):
            return term(p)
        op, name = levels[i]
        left, p = level(i + 1, p)
        while p < len(tokens) and tokens[p] == op:
            right, p = level(i + 1, p + 1)
            left = (name, left, right)
        return left, p

    return level(0, pos)


def parse_schema(text: str) -> dict:
    text = _strip_comments(text)
    if re.search(r"\bcaveat\s+\w+\s*\(", text):
        raise SchemaError("unsupported: caveats")
    defs: dict[str, Definition] = {}
    text = re.sub(r"^\s*use\s+\w+\s*$", "", text, flags=re.M)
    for m in re.finditer(r"definition\s+([\w/]+)\s*\{(.*?)\}", text, flags=re.S):
        d = Definition(m.group(1))
        if d.name in defs:
            raise SchemaError("duplicate definition")
        body = m.group(2)
        for line in re.finditer(r"(relation|permission)\s+(\w+)\s*([:=])\s*([^\n]*?)(?=\s*(?:relation\s|permission\s|$))", body, flags=re.S):
            kind, name, sep, rest = line.groups()
            if name in d.members:
                raise SchemaError("duplicate member")
            if kind == "relation":
                if sep != ":":
                    raise SchemaError("relation needs ':'")
                allowed = []
                for alt in rest.split("|"):
                    alt = alt.strip()
                    mm = re.fullmatch(r"([\w/]+)(?:#(\w+)|(:\*))?(\s+with\s+expiration)?", alt)
                    if not mm:
                        raise SchemaError(f"unsupported subject reference {alt!r}")
                    allowed.append((mm.group(1), "*" if mm.group(3) else mm.group(2), bool(mm.group(4))))
                d.members[name] = Relation(name, allowed)
            else:
                if sep != "=":
                    raise SchemaError("permission needs '='")
                toks = re.findall(r"->|[A-Za-z_][\w/]*|[()+&\-.]", rest)
                if re.sub(r"->|[A-Za-z_][\w/]*|[()+&\-.]|\s", "", rest):
                    raise SchemaError(f"unexpected characters in permission expression {rest!r}")
                expr, pos = _parse_expr(toks, 0)
                if pos != len(toks):
                    raise SchemaError("trailing tokens in permission expression")
                d.members[name] = Permission(name, expr)
        defs[d.name] = d
    leftover = re.sub(r"definition\s+[\w/]+\s*\{.*?\}", "", text, flags=re.S).strip()
    if leftover:
        raise SchemaError(f"unparsed schema text: {leftover[:40]!r}")
    for d in defs.values():
        for mem in d.members.values():
            if isinstance(mem, Relation):
                for (st, sr, _e) in mem.allowed:
                    if st not in defs or (sr and sr != "*" and sr not in defs[st].members):
                        raise SchemaError(f"unknown subject reference {st}#{sr}")
            else:
                _validate(defs, d, mem.expr)
    return defs


def _validate(defs, d, e):
    if e[0] in ("union", "inter", "excl"):
        _validate(defs, d, e[1])
        _validate(defs, d, e[2])
    elif e[0] == "ref" and e[1] not in d.members:
        raise SchemaError(f"unknown reference {e[1]}")
    elif e[0] in ("arrow", "arrow_all"):
        ts = d.members.get(e[1])
        if not isinstance(ts, Relation):
            raise SchemaError(f"arrow over non-relation {e[1]}")
        if any(a[1] == "*" for a in ts.allowed):
            raise SchemaError(f"arrow over {e[1]}, which allows wildcard subjects")
        if e[0] == "arrow_all" and any(e[2] not in defs[a[0]].members for a in ts.allowed):  # fails closed (see oracle/acl_oracle.c)
            raise SchemaError(f"{e[1]}.all({e[2]}) over a subject type without {e[2]}")


# intersection: an empty operand decides (NO), then an error, then HAS; exclusion with a HAS base: by the subtracted operand
_INTER = {(a, b): (NO if NO in (a, b) else ERR if ERR in (a, b) else HAS) for a in (NO, HAS, ERR) for b in (NO, HAS, ERR)}
_MINUS = {NO: HAS, HAS: NO, ERR: ERR}


# ---- API validation (EXTERNAL, unverified: the `validate` rules of authzed.api.v1, see oracle/acl_oracle.c): whole-request InvalidArgument
_ID = re.compile(r"[a-zA-Z0-9/_|\-=+]{1,1024}\Z")
_REL = re.compile(r"[a-z][a-z0-9_]{1,62}[a-z0-9]\Z")
_TYPE = re.compile(r"([a-z][a-z0-9_]{1,61}[a-z0-9]/)*[a-z][a-z0-9_]{1,62}[a-z0-9]\Z")


class InvalidArgument(ValueError):
    pass


class LookupFailed(Exception):
    """a candidate's check erred: the whole LookupResources call fails (code 100 = max depth exceeded)"""
    code = 100


class PyOracle:
    """Relationship store + evaluator.  Tuples are 6-tuples of strings
    (rtype, rid, rel, stype, sid, srel) with srel == '' for no subject relation."""

    def __init__(self, schema: str):
        self.defs = parse_schema(schema)
        self.rows: dict = {}  # (rtype, rid, rel) -> {(stype, sid, srel): expires}
        self.now = 0
        self.relaxed = False         # evaluate the POSITIVE relaxation of the schema (lookup_resources: which resources are candidates)
        self.lenient_lookup = False  # True: a candidate whose check errs is dropped instead of failing the lookup

    # -- writes (TOUCH semantics; the C oracle covers CREATE/preconditions)
    def touch(self, rtype, rid, rel, stype, sid, srel="", expires=0):
        self.validate(rtype, rid, rel, stype, sid, srel, star_subject=True)
        mem = self.defs[rtype].members[rel]
        assert isinstance(mem, Relation)
        if sid == "*":  # `T:*`: its own allowed form, stored as the subject (T, "*", "")
            assert srel == "" and any(a[0] == stype and a[1] == "*" for a in mem.allowed), "wildcard not allowed"
        else:
            assert any(a[0] == stype and a[1] != "*" and (a[1] or "") == srel for a in mem.allowed), "subject not allowed"
        self.rows.setdefault((rtype, rid, rel), {})[(stype, sid, srel)] = expires

    def delete(self, rtype, rid, rel, stype, sid, srel=""):
        self.rows.get((rtype, rid, rel), {}).pop((stype, sid, srel), None)

    def _subjects(self, rtype, rid, rel):
        return [s for s, exp in self.rows.get((rtype, rid, rel), {}).items() if exp == 0 or exp > self.now]

    # -- evaluation: one call == one dispatch
    def _check(self, rtype, rid, rel, subject, depth):
        # pure in (state, depth) for one top-level call: memoised so that cyclic
        # data stays polynomial (answers are unchanged by the memo)
        key = (rtype, rid, rel, depth)
        if key not in self._memo:
            self._memo[key] = self._check_body(rtype, rid, rel, subject, depth)
        return self._memo[key]

    def _check_body(self, rtype, rid, rel, subject, depth):
        if depth <= 0:
            return ERR
        if subject == (rtype, rid, rel):
            return HAS
        mem = self.defs[rtype].members[rel]
        results = []
        if isinstance(mem, Relation):
            subs = self._subjects(rtype, rid, rel)
            if subject in subs:
                return HAS
            if subject[2] == "" and (subject[0], "*", "") in subs:
                return HAS  # a wildcard covers every plain subject of its type
            for (st, sid, sr) in subs:
                if sr:
                    results.append(self._check(st, sid, sr, subject, depth - 1))
        else:
            results.append(self._eval(rtype, rid, mem.expr, subject, depth))
        return HAS if HAS in results else ERR if ERR in results else NO

    def _eval(self, rtype, rid, e, subject, depth):
        if e[0] == "nil":
            return NO
        if self.relaxed:  # candidates: `a & b` as a + b, `a - b` as a, a.all(b) as a->b
            if e[0] == "excl":
                return self._eval(rtype, rid, e[1], subject, depth)
            if e[0] in ("inter", "arrow_all"):
                e = ("union" if e[0] == "inter" else "arrow", e[1], e[2])
        if e[0] == "inter":
            a = self._eval(rtype, rid, e[1], subject, depth)
            b = NO if a == NO else self._eval(rtype, rid, e[2], subject, depth)
            return _INTER[a, b]
        if e[0] == "excl":
            a = self._eval(rtype, rid, e[1], subject, depth)
            return a if a != HAS else _MINUS[self._eval(rtype, rid, e[2], subject, depth)]
        if e[0] == "arrow_all":  # intersection arrow (EXTERNAL, unverified): every subject of the tupleset must hold the permission, and there must be one
            rs = [self._check(st, sid, e[2], subject, depth - 1) for (st, sid, _sr) in self._subjects(rtype, rid, e[1]) if e[2] in self.defs[st].members]
            return NO if not rs or NO in rs else ERR if ERR in rs else HAS
        if e[0] == "union":
            rs = [self._eval(rtype, rid, e[1], subject, depth), self._eval(rtype, rid, e[2], subject, depth)]
        elif e[0] == "ref":
            rs = [self._check(rtype, rid, e[1], subject, depth - 1)]
        else:  # arrow
            rs = []
            for (st, sid, _sr) in self._subjects(rtype, rid, e[1]):
                if e[2] in self.defs[st].members:
                    rs.append(self._check(st, sid, e[2], subject, depth - 1))
        return HAS if HAS in rs else ERR if ERR in rs else NO

    def validate(self, rtype, rid, rel, stype, sid, srel="", star_subject=False):
        """raises InvalidArgument for a request the API's validation refuses (declared names pass whatever their spelling)"""
        d, sd = self.defs.get(rtype), self.defs.get(stype)
        ok = (d is not None or (len(rtype) <= 128 and _TYPE.match(rtype))) and (sd is not None or (len(stype) <= 128 and _TYPE.match(stype)))
        ok = ok and ((d is not None and rel in d.members) or _REL.match(rel))
        ok = ok and (srel in ("", "...") or (sd is not None and srel in sd.members) or _REL.match(srel))
        ok = ok and (rid is None or _ID.match(rid)) and (_ID.match(sid) or (star_subject and sid == "*"))
        if not ok:
            raise InvalidArgument((rtype, rid, rel, stype, sid, srel))

    def check(self, rtype, rid, perm, stype, sid, srel=""):
        """returns 'HAS' | 'NO' | 'ERR' (depth) ; raises InvalidArgument for an ill-formed request, KeyError for unknown type/relation."""
        self.validate(rtype, rid, perm, stype, sid, srel)
        if perm not in self.defs[rtype].members:
            raise KeyError(perm)
        if srel and srel not in self.defs[stype].members:
            raise KeyError(srel)
        self._memo = {}
        return self._check(rtype, rid, perm, (stype, sid, srel), MAX_DEPTH)

    def lookup_resources(self, rtype, perm, stype, sid, srel=""):
        self.validate(rtype, None, perm, stype, sid, srel)
        ids = {k[1] for k in self.rows if k[0] == rtype}
        if stype == rtype:
            ids.add(sid)
        # A resource whose check ERRS fails the whole lookup iff it is a CANDIDATE -- iff the positive relaxation of the schema grants it, i.e. a
        # reverse reachability walk from the subject finds it (EXTERNAL, unverified; reference pkg/authz/lookups.go:75-83: the stream ends at the
        # first Recv error).  Resources on cycles / long chains the subject has nothing to do with stay out silently.
        out = set()
        for i in ids:
            r = self.check(rtype, i, perm, stype, sid, srel)
            if r == HAS:
                out.add(i)
            elif r == ERR and not self.lenient_lookup:
                self.relaxed = True
                try:
                    cand = self.check(rtype, i, perm, stype, sid, srel) == HAS
                finally:
                    self.relaxed = False
                if cand:
                    raise LookupFailed(f"LookupResources: max depth exceeded while checking candidate {rtype}:{i}")
        return out
