/*
 * acl_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the permission arithmetic that the reference reaches
 * through v1.PermissionsServiceClient:
 *   - CheckBulkPermissions  (reference call sites pkg/authz/check.go:48,
 *                            pkg/authz/postfilter.go:134)
 *   - CheckPermission       (pkg/authz/watch.go:50)
 *   - LookupResources       (pkg/authz/lookups.go:65)
 *   - WriteRelationships / ReadRelationships / DeleteRelationships
 *                           (pkg/authz/distributedtx/activity.go:60,107;
 *                            e2e/util_test.go:27,66)
 * and that is actually implemented in the THIRD-PARTY module
 *   github.com/authzed/spicedb v1.53.1-0.20260609214739-87482ed4fbea (go.mod:9)
 * which is NOT present under /root/reference and cannot be built here (no Go
 * toolchain, no module cache, no network).  The evaluation rules below restate
 * SpiceDB's published behaviour for the schema subset the reference uses
 * (pkg/spicedb/bootstrap.yaml:1-40; e2e rules): direct relations, userset
 * subjects (`group#member`), union `+`, `nil`, arrows `a->b`, dispatch depth
 * limit 50 (pkg/spicedb/spicedb.go:34), relationship expiration
 * (pkg/spicedb/spicedb.go:60), and -- round 4, because the reference boots ARBITRARY
 * schemas (pkg/spicedb/spicedb.go:19-24, pkg/proxy/options.go:313-316,
 * e2e/embedded_integration_test.go:34-250) -- intersection `&`, exclusion `-`,
 * wildcard subjects `type:*` and intersection arrows `a.all(b)` (every subject of `a`
 * must hold `b`, and there must be one; eval_expr EX_ARROW_ALL).  Caveats are REJECTED
 * at schema load.
 *
 * Intersection / exclusion (EXTERNAL: SpiceDB internal/graph/check.go `all` /
 * `difference`, restated from memory, unverified):
 *   - operator precedence of the schema language, loosest to tightest: `-`, `&`, `+`
 *     (`a + b - c` is `(a + b) - c`, `a - b + c` is `a - (b + c)`); same-operator
 *     chains associate to the left; parentheses override;
 *   - neither operator dispatches: their operands (references, arrows) do, exactly as
 *     under a union, so the depth limit counts the same dispatches;
 *   - results are three-valued per item.  SpiceDB evaluates operands concurrently and
 *     takes the first DECISIVE result, so which of an error and a decisive result wins
 *     is a race there; this restatement fixes the order a short-circuiting evaluator
 *     would most often see: union HAS > ERR > NO; intersection NO > ERR > HAS;
 *     exclusion: base ERR -> ERR, base NO -> NO, then subtracted ERR -> ERR,
 *     subtracted HAS -> NO, else HAS.  Generators keep data acyclic so that ERR only
 *     arises where a test asks for it.
 * Wildcards (EXTERNAL, same caveat): a relationship `res#rel@T:*` makes every subject
 * `T:x` WITHOUT a subject relation a member of res#rel; it is never a userset, arrows
 * may not walk a relation that allows wildcards, `*` cannot carry a relation.
 *
 * PARITY STATUS: the oracle is pinned against every golden vector the
 * reference's own tests hold for this path (see
 * tests/test_oracle_golden.py).  Arrows, nested usersets and the depth limit
 * are NOT pinned by any reference test ("parity unpinned" for those shapes);
 * they are cross-checked only against the independent Python restatement in
 * oracle/pyoracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product (libaclgpu.so) never does.
 *
 * Structure deliberately mirrors SpiceDB's recursive dispatch (one C call per
 * dispatched sub-check, depth decremented per dispatch) and shares NO code with
 * the engine: own schema parser, own interner, own sorted tuple index.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <stdarg.h>

#define ORC_MAX_DEPTH 50 /* pkg/spicedb/spicedb.go:34 WithDispatchMaxDepth(50) */

/* authzed.api.v1 enum values (EXTERNAL: authzed-go v1.10.0, go.mod:6) */
enum { ORC_PERM_UNSPEC = 0, ORC_PERM_NO = 1, ORC_PERM_HAS = 2, ORC_PERM_COND = 3 };
enum { ORC_OP_CREATE = 1, ORC_OP_TOUCH = 2, ORC_OP_DELETE = 3 };
enum { ORC_PRE_MUST_NOT_MATCH = 1, ORC_PRE_MUST_MATCH = 2 };
/* error codes = gRPC status codes the reference inspects */
enum {
    ORC_OK = 0,
    ORC_ERR_INVALID_ARGUMENT = 3,    /* codes.InvalidArgument */
    ORC_ERR_ALREADY_EXISTS = 6,      /* CREATE of an existing tuple */
    ORC_ERR_FAILED_PRECONDITION = 9, /* precondition failed / unknown type or relation */
    ORC_ERR_DEPTH = 100              /* max dispatch depth exceeded (per-item error) */
};

#define ELLIPSIS 0xFFFFu
#define WILDCARD 0xFFFEu /* tuple_t.srel of `T:*` subjects (subj = the interned id of "*", never compared) */

/* ------------------------------------------------------------------ schema */
enum { EX_UNION, EX_REF, EX_ARROW, EX_NIL, EX_INTERSECT, EX_EXCLUDE, EX_ARROW_ALL /* a.all(b): the intersection arrow */ };
typedef struct expr {
    int kind;
    struct expr *l, *r; /* union, intersection; exclusion: l minus r */
    char *a, *b;        /* ref: a ; arrow: a->b */
} expr_t;

typedef struct {
    int stype;      /* subject type index */
    unsigned srel;  /* relation index in subject type, ELLIPSIS, or WILDCARD (`T:*`) */
    int expiring;   /* `with expiration` */
} allowed_t;

typedef struct {
    char *name;
    int is_perm;
    allowed_t *allowed;
    int nallowed;
    /* unresolved allowed refs (names) until all types are parsed */
    char **a_type, **a_rel;
    int *a_exp, *a_wild;
    expr_t *expr;
} rel_t;

typedef struct {
    char **strs;
    uint32_t n, cap;
    uint32_t *ht;
    uint32_t htcap;
} strtab_t;

typedef struct {
    char *name;
    rel_t *rels;
    int nrels;
    strtab_t objs;
} type_t;

typedef struct {
    uint16_t rtype, rel;
    uint32_t res;
    uint16_t stype, srel;
    uint32_t subj;
    int64_t expires; /* unix seconds, 0 = never */
} tuple_t;

typedef struct orc {
    type_t *types;
    int ntypes;
    tuple_t *tup;
    size_t ntup, captup;
    int sorted;
    int64_t now;      /* evaluation time for expiration, unix seconds */
    uint64_t revision;
    char err[512];
    /* scratch for lookups */
    uint32_t *lr_ids;
    size_t lr_n, lr_cap;
    /* work counters of the last check call */
    uint64_t cnt_dispatch, cnt_rows, cnt_edges;
    /* per-call memo of check_rel(state, depth) -- see memo_get() */
    uint64_t *memo_k;
    uint8_t *memo_v;
    size_t memo_cap, memo_n;
    uint64_t memo_base; /* cnt_dispatch at the start of the current top-level check */
    /* LookupResources: the POSITIVE relaxation of the schema (`a & b` -> a + b, `a - b` -> a, a.all(b) -> a->b) decides which resources are
     * CANDIDATES -- what a reverse reachability walk from the subject finds; see orc_lookup_ids() */
    int relaxed;
    int lenient_lookup; /* 1: a candidate whose Check errs is dropped instead of failing the lookup (orc_set_lenient_lookup) */
    int lookup_err;     /* code of the last failed orc_lookup_ids() */
    /* the tuned evaluator's row index (orc_tuned_*): first tuple of (type, rel, resource id) */
    uint32_t **tn_rowptr; /* [ntypes * maxrels] -> [maxid + 2], NULL where the relation has no tuples */
    uint32_t *tn_maxid;   /* [ntypes * maxrels] */
    int tn_maxrels, tn_ready;
} orc_t;

static void seterr(orc_t *o, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(o->err, sizeof o->err, fmt, ap);
    va_end(ap);
}

/* ------------------------------------------------------------ string table */
static uint64_t fnv1a(const char *s) {
    uint64_t h = 1469598103934665603ull;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ull; }
    return h;
}
static void st_rehash(strtab_t *t, uint32_t ncap) {
    uint32_t *nh = malloc(sizeof(uint32_t) * ncap);
    memset(nh, 0xFF, sizeof(uint32_t) * ncap);
    for (uint32_t i = 0; i < t->n; i++) {
        uint64_t h = fnv1a(t->strs[i]) & (ncap - 1);
        while (nh[h] != 0xFFFFFFFFu) h = (h + 1) & (ncap - 1);
        nh[h] = i;
    }
    free(t->ht);
    t->ht = nh;
    t->htcap = ncap;
}
static uint32_t st_find(const strtab_t *t, const char *s) {
    if (!t->htcap) return 0xFFFFFFFFu;
    uint64_t h = fnv1a(s) & (t->htcap - 1);
    while (t->ht[h] != 0xFFFFFFFFu) {
        if (strcmp(t->strs[t->ht[h]], s) == 0) return t->ht[h];
        h = (h + 1) & (t->htcap - 1);
    }
    return 0xFFFFFFFFu;
}
static uint32_t st_intern(strtab_t *t, const char *s) {
    uint32_t f = st_find(t, s);
    if (f != 0xFFFFFFFFu) return f;
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->strs = realloc(t->strs, sizeof(char *) * t->cap);
    }
    t->strs[t->n] = strdup(s);
    t->n++;
    if (t->n * 2 > t->htcap) st_rehash(t, t->htcap ? t->htcap * 2 : 128);
    else {
        uint64_t h = fnv1a(s) & (t->htcap - 1);
        while (t->ht[h] != 0xFFFFFFFFu) h = (h + 1) & (t->htcap - 1);
        t->ht[h] = t->n - 1;
    }
    return t->n - 1;
}
static void st_free(strtab_t *t) {
    for (uint32_t i = 0; i < t->n; i++) free(t->strs[i]);
    free(t->strs);
    free(t->ht);
}

/* ------------------------------------------------------------------ lexer */
typedef struct {
    const char *p;
    int tok; /* 0 eof, 'i' ident, or punctuation char; 'A' for "->" */
    char text[256];
    orc_t *o;
    int failed;
} lex_t;

static void lx_next(lex_t *L) {
    const char *p = L->p;
    for (;;) {
        while (*p && isspace((unsigned char)*p)) p++;
        if (p[0] == '/' && p[1] == '/') { while (*p && *p != '\n') p++; continue; }
        if (p[0] == '/' && p[1] == '*') {
            p += 2;
            while (*p && !(p[0] == '*' && p[1] == '/')) p++;
            if (*p) p += 2;
            continue;
        }
        break;
    }
    if (!*p) { L->tok = 0; L->p = p; return; }
    if (isalpha((unsigned char)*p) || *p == '_') {
        int n = 0;
        /* identifiers may carry a tenant prefix: foo/bar */
        while ((isalnum((unsigned char)*p) || *p == '_' || *p == '/') && n < 255) L->text[n++] = *p++;
        L->text[n] = 0;
        L->tok = 'i';
        L->p = p;
        return;
    }
    if (p[0] == '-' && p[1] == '>') { L->tok = 'A'; L->p = p + 2; return; }
    L->tok = *p;
    L->p = p + 1;
}
static int lx_fail(lex_t *L, const char *msg) {
    if (!L->failed) seterr(L->o, "schema: %s near '%.20s'", msg, L->p);
    L->failed = 1;
    return 0;
}
static int lx_expect(lex_t *L, int tok, const char *what) {
    if (L->tok != tok) return lx_fail(L, what);
    lx_next(L);
    return 1;
}

static expr_t *ex_new(int kind) {
    expr_t *e = calloc(1, sizeof *e);
    e->kind = kind;
    return e;
}
static void ex_free(expr_t *e) {
    if (!e) return;
    ex_free(e->l); ex_free(e->r);
    free(e->a); free(e->b);
    free(e);
}
static expr_t *parse_expr(lex_t *L);
static expr_t *parse_term(lex_t *L) {
    if (L->tok == '(') {
        lx_next(L);
        expr_t *e = parse_expr(L);
        if (!lx_expect(L, ')', "expected ')'")) { ex_free(e); return NULL; }
        return e;
    }
    if (L->tok != 'i') { lx_fail(L, "expected identifier in permission expression"); return NULL; }
    if (strcmp(L->text, "nil") == 0) { lx_next(L); return ex_new(EX_NIL); }
    char a[256];
    strcpy(a, L->text);
    lx_next(L);
    if (L->tok == 'A') {
        lx_next(L);
        if (L->tok != 'i') { lx_fail(L, "expected identifier after '->'"); return NULL; }
        expr_t *e = ex_new(EX_ARROW);
        e->a = strdup(a); e->b = strdup(L->text);
        lx_next(L);
        return e;
    }
    if (L->tok == '.') { /* a.any(b) == a->b ; a.all(b): b must hold on EVERY subject of a (and there must be one) */
        lx_next(L);
        if (L->tok != 'i') { lx_fail(L, "expected any/all"); return NULL; }
        const int all = strcmp(L->text, "all") == 0;
        if (!all && strcmp(L->text, "any") != 0) { lx_fail(L, "unsupported arrow function (any / all)"); return NULL; }
        lx_next(L);
        if (!lx_expect(L, '(', "expected '('")) return NULL;
        if (L->tok != 'i') { lx_fail(L, "expected identifier"); return NULL; }
        expr_t *e = ex_new(all ? EX_ARROW_ALL : EX_ARROW);
        e->a = strdup(a); e->b = strdup(L->text);
        lx_next(L);
        if (!lx_expect(L, ')', "expected ')'")) { ex_free(e); return NULL; }
        return e;
    }
    expr_t *e = ex_new(EX_REF);
    e->a = strdup(a);
    return e;
}
/* one binary level: operands from `sub`, joined by operator character `op` into nodes of `kind`, left-associative */
static expr_t *parse_level(lex_t *L, int op, int kind, expr_t *(*sub)(lex_t *)) {
    expr_t *l = sub(L);
    if (!l) return NULL;
    while (L->tok == op) {
        lx_next(L);
        expr_t *r = sub(L);
        if (!r) { ex_free(l); return NULL; }
        expr_t *u = ex_new(kind);
        u->l = l; u->r = r;
        l = u;
    }
    return l;
}
/* precedence, loosest first: exclusion, intersection, union (EXTERNAL, see the header) */
static expr_t *parse_union(lex_t *L) { return parse_level(L, '+', EX_UNION, parse_term); }
static expr_t *parse_inter(lex_t *L) { return parse_level(L, '&', EX_INTERSECT, parse_union); }
static expr_t *parse_expr(lex_t *L) { return parse_level(L, '-', EX_EXCLUDE, parse_inter); }

static int type_index(orc_t *o, const char *name) {
    for (int i = 0; i < o->ntypes; i++)
        if (strcmp(o->types[i].name, name) == 0) return i;
    return -1;
}
static int rel_index(const type_t *t, const char *name) {
    for (int i = 0; i < t->nrels; i++)
        if (strcmp(t->rels[i].name, name) == 0) return i;
    return -1;
}

static int parse_schema(orc_t *o, const char *text) {
    lex_t L = {.p = text, .o = o};
    lx_next(&L);
    while (L.tok && !L.failed) {
        if (L.tok != 'i') return lx_fail(&L, "expected 'definition'");
        if (strcmp(L.text, "use") == 0) { /* `use expiration` etc. */
            lx_next(&L);
            if (L.tok != 'i') return lx_fail(&L, "expected feature name after 'use'");
            lx_next(&L);
            continue;
        }
        if (strcmp(L.text, "caveat") == 0) return lx_fail(&L, "unsupported: caveats");
        if (strcmp(L.text, "definition") != 0) return lx_fail(&L, "expected 'definition'");
        lx_next(&L);
        if (L.tok != 'i') return lx_fail(&L, "expected definition name");
        if (type_index(o, L.text) >= 0) return lx_fail(&L, "duplicate definition");
        o->types = realloc(o->types, sizeof(type_t) * (o->ntypes + 1));
        type_t *t = &o->types[o->ntypes++];
        memset(t, 0, sizeof *t);
        t->name = strdup(L.text);
        lx_next(&L);
        if (!lx_expect(&L, '{', "expected '{'")) return 0;
        while (L.tok == 'i' && !L.failed) {
            int is_perm = strcmp(L.text, "permission") == 0;
            if (!is_perm && strcmp(L.text, "relation") != 0) return lx_fail(&L, "expected relation/permission");
            lx_next(&L);
            if (L.tok != 'i') return lx_fail(&L, "expected name");
            if (rel_index(t, L.text) >= 0) return lx_fail(&L, "duplicate relation/permission");
            t->rels = realloc(t->rels, sizeof(rel_t) * (t->nrels + 1));
            rel_t *r = &t->rels[t->nrels++];
            memset(r, 0, sizeof *r);
            r->name = strdup(L.text);
            r->is_perm = is_perm;
            lx_next(&L);
            if (is_perm) {
                if (!lx_expect(&L, '=', "expected '='")) return 0;
                r->expr = parse_expr(&L);
                if (!r->expr) return 0;
            } else {
                if (!lx_expect(&L, ':', "expected ':'")) return 0;
                for (;;) {
                    if (L.tok != 'i') return lx_fail(&L, "expected subject type");
                    int k = r->nallowed++;
                    r->a_type = realloc(r->a_type, sizeof(char *) * r->nallowed);
                    r->a_rel = realloc(r->a_rel, sizeof(char *) * r->nallowed);
                    r->a_exp = realloc(r->a_exp, sizeof(int) * r->nallowed);
                    r->a_wild = realloc(r->a_wild, sizeof(int) * r->nallowed);
                    r->a_type[k] = strdup(L.text);
                    r->a_rel[k] = NULL;
                    r->a_exp[k] = 0;
                    r->a_wild[k] = 0;
                    lx_next(&L);
                    if (L.tok == '#') {
                        lx_next(&L);
                        if (L.tok != 'i') return lx_fail(&L, "expected relation after '#'");
                        r->a_rel[k] = strdup(L.text);
                        lx_next(&L);
                    } else if (L.tok == ':') { /* `T:*` */
                        lx_next(&L);
                        if (L.tok != '*') return lx_fail(&L, "expected '*' after ':'");
                        r->a_wild[k] = 1;
                        lx_next(&L);
                    }
                    if (L.tok == 'i' && strcmp(L.text, "with") == 0) {
                        lx_next(&L);
                        if (L.tok != 'i' || strcmp(L.text, "expiration") != 0)
                            return lx_fail(&L, "unsupported: caveated relations");
                        r->a_exp[k] = 1;
                        lx_next(&L);
                        if (L.tok == 'i' && strcmp(L.text, "and") == 0) return lx_fail(&L, "unsupported: caveated relations");
                    }
                    if (L.tok == '|') { lx_next(&L); continue; }
                    break;
                }
            }
        }
        if (!lx_expect(&L, '}', "expected '}'")) return 0;
    }
    if (L.failed) return 0;
    /* resolve allowed subject references */
    for (int ti = 0; ti < o->ntypes; ti++) {
        type_t *t = &o->types[ti];
        for (int ri = 0; ri < t->nrels; ri++) {
            rel_t *r = &t->rels[ri];
            if (r->is_perm) continue;
            r->allowed = calloc(r->nallowed ? r->nallowed : 1, sizeof(allowed_t));
            for (int k = 0; k < r->nallowed; k++) {
                int st = type_index(o, r->a_type[k]);
                if (st < 0) { seterr(o, "schema: unknown subject type '%s' in %s#%s", r->a_type[k], t->name, r->name); return 0; }
                unsigned sr = r->a_wild[k] ? WILDCARD : ELLIPSIS;
                if (r->a_rel[k]) {
                    int x = rel_index(&o->types[st], r->a_rel[k]);
                    if (x < 0) { seterr(o, "schema: unknown relation '%s#%s'", r->a_type[k], r->a_rel[k]); return 0; }
                    sr = (unsigned)x;
                }
                r->allowed[k].stype = st;
                r->allowed[k].srel = sr;
                r->allowed[k].expiring = r->a_exp[k];
            }
        }
    }
    /* validate permission expressions: refs must exist; arrow tuplesets must be relations */
    for (int ti = 0; ti < o->ntypes; ti++) {
        type_t *t = &o->types[ti];
        for (int ri = 0; ri < t->nrels; ri++) {
            rel_t *r = &t->rels[ri];
            if (!r->is_perm) continue;
            expr_t *stack[256];
            int sp = 0;
            stack[sp++] = r->expr;
            while (sp) {
                expr_t *e = stack[--sp];
                if (e->kind == EX_UNION || e->kind == EX_INTERSECT || e->kind == EX_EXCLUDE) { stack[sp++] = e->l; stack[sp++] = e->r; }
                else if (e->kind == EX_REF) {
                    if (rel_index(t, e->a) < 0) { seterr(o, "schema: %s#%s references unknown '%s'", t->name, r->name, e->a); return 0; }
                } else if (e->kind == EX_ARROW || e->kind == EX_ARROW_ALL) {
                    int x = rel_index(t, e->a);
                    if (x < 0 || t->rels[x].is_perm) { seterr(o, "schema: %s#%s arrow over non-relation '%s'", t->name, r->name, e->a); return 0; }
                    for (int k = 0; k < t->rels[x].nallowed; k++)
                        if (t->rels[x].allowed[k].srel == WILDCARD) { seterr(o, "schema: %s#%s arrow over '%s', which allows wildcard subjects", t->name, r->name, e->a); return 0; }
                    /* a.all(b) FAILS CLOSED: "every subject of a holds b" is not defined here for a subject whose type has no b -- a plain
                     * arrow skips such subjects, an intersection arrow that skipped them could grant what the real engine denies: refused at load */
                    if (e->kind == EX_ARROW_ALL)
                        for (int k = 0; k < t->rels[x].nallowed; k++)
                            if (rel_index(&o->types[t->rels[x].allowed[k].stype], e->b) < 0) {
                                seterr(o, "schema: %s#%s: %s.all(%s) over a subject type without '%s'", t->name, r->name, e->a, e->b, e->b);
                                return 0;
                            }
                }
            }
        }
    }
    return 1;
}

/* ---------------------------------------------------------- API validation
 * EXTERNAL, UNVERIFIED (authzed.api.v1 `validate` rules, authzed-go v1.10.0, go.mod:6; restated from memory): a request with an
 * ill-formed field is InvalidArgument as a whole, BEFORE the schema is consulted.
 *   object type     ^([a-z][a-z0-9_]{1,61}[a-z0-9]/)*[a-z][a-z0-9_]{1,62}[a-z0-9]$   <= 128 bytes
 *   relation        ^[a-z][a-z0-9_]{1,62}[a-z0-9]$                                    <= 64 bytes
 *   object id       ^[a-zA-Z0-9/_|\-=+]+$                                            <= 1024 bytes
 *   `*`             only as the subject id of a relationship / relationship filter, without a relation
 * A name the loaded schema DECLARES is accepted whatever its spelling (this restatement's schema parser takes names the real
 * compiler refuses; for schemas the real engine accepts the two readings coincide).  Reference-held vector: the empty
 * CheckPermissionRequest, pkg/proxy/options_test.go:101-102. */
static int ok_id(const char *s) {
    size_t n = strlen(s);
    if (!n || n > 1024) return 0;
    for (; *s; s++)
        if (!(isalnum((unsigned char)*s) || strchr("/_|-=+", *s))) return 0;
    return 1;
}
static int ok_segment(const char *s, size_t n, size_t maxlen) {
    if (n < 3 || n > maxlen) return 0;
    if (!(s[0] >= 'a' && s[0] <= 'z')) return 0;
    if (!((s[n - 1] >= 'a' && s[n - 1] <= 'z') || isdigit((unsigned char)s[n - 1]))) return 0;
    for (size_t i = 1; i + 1 < n; i++)
        if (!((s[i] >= 'a' && s[i] <= 'z') || isdigit((unsigned char)s[i]) || s[i] == '_')) return 0;
    return 1;
}
static int ok_relname(const char *s) { return ok_segment(s, strlen(s), 64); }
static int ok_typename(const char *s) {
    if (strlen(s) > 128) return 0;
    for (;;) {
        const char *slash = strchr(s, '/');
        if (!slash) return ok_segment(s, strlen(s), 64);
        if (!ok_segment(s, (size_t)(slash - s), 63)) return 0;
        s = slash + 1;
    }
}
static int rel_index(const type_t *t, const char *name);
static int type_index(orc_t *o, const char *name);
/* names of a request: declared or well-formed; srel may be NULL / "" / "..." */
static int names_wellformed(orc_t *o, const char *rtype, const char *rel, const char *stype, const char *srel) {
    int rt = type_index(o, rtype), st = stype ? type_index(o, stype) : -1;
    if (rt < 0 && !ok_typename(rtype)) return 0;
    if (stype && st < 0 && !ok_typename(stype)) return 0;
    if (rel && *rel && (rt < 0 || rel_index(&o->types[rt], rel) < 0) && !ok_relname(rel)) return 0;
    if (srel && *srel && strcmp(srel, "...") != 0 && (st < 0 || rel_index(&o->types[st], srel) < 0) && !ok_relname(srel)) return 0;
    return 1;
}

/* --------------------------------------------------------------- tuple index */
static int tup_cmp(const void *pa, const void *pb) {
    const tuple_t *a = pa, *b = pb;
    if (a->rtype != b->rtype) return a->rtype < b->rtype ? -1 : 1;
    if (a->rel != b->rel) return a->rel < b->rel ? -1 : 1;
    if (a->res != b->res) return a->res < b->res ? -1 : 1;
    if (a->stype != b->stype) return a->stype < b->stype ? -1 : 1;
    if (a->srel != b->srel) return a->srel < b->srel ? -1 : 1;
    if (a->subj != b->subj) return a->subj < b->subj ? -1 : 1;
    return 0;
}
static void tn_drop(orc_t *o) { /* the tuned evaluator's row index follows the tuple array: rebuilt on the next orc_tuned_* call */
    if (o->tn_rowptr) {
        for (size_t k = 0; k < (size_t)o->ntypes * (size_t)o->tn_maxrels; k++) free(o->tn_rowptr[k]);
        free(o->tn_rowptr);
        free(o->tn_maxid);
    }
    o->tn_rowptr = NULL;
    o->tn_maxid = NULL;
    o->tn_ready = 0;
}
static void ensure_sorted(orc_t *o) {
    if (o->sorted) return;
    tn_drop(o);
    qsort(o->tup, o->ntup, sizeof(tuple_t), tup_cmp);
    /* dedupe (TOUCH semantics for bulk loads) */
    size_t w = 0;
    for (size_t i = 0; i < o->ntup; i++) {
        if (w && tup_cmp(&o->tup[w - 1], &o->tup[i]) == 0) { o->tup[w - 1] = o->tup[i]; continue; }
        o->tup[w++] = o->tup[i];
    }
    o->ntup = w;
    o->sorted = 1;
}
/* first index with tup >= key */
static size_t lower_bound(const orc_t *o, const tuple_t *key) {
    size_t lo = 0, hi = o->ntup;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (tup_cmp(&o->tup[mid], key) < 0) lo = mid + 1; else hi = mid;
    }
    return lo;
}
static int tup_live(const orc_t *o, const tuple_t *t) { return t->expires == 0 || t->expires > o->now; }

/* range of tuples for (rtype, rel, res): [*lo, *hi) */
static void row_range(const orc_t *o, int rtype, int rel, uint32_t res, size_t *lo, size_t *hi) {
    tuple_t k = {.rtype = (uint16_t)rtype, .rel = (uint16_t)rel, .res = res, .stype = 0, .srel = 0, .subj = 0};
    *lo = lower_bound(o, &k);
    /* no real tuple has stype 0xFFFF, so this key sorts after the whole row */
    k.stype = 0xFFFF; k.srel = 0xFFFF; k.subj = 0xFFFFFFFFu;
    *hi = lower_bound(o, &k);
}
/* index of the tuple with this exact subject inside row [lo,hi), or hi */
static size_t row_find(const orc_t *o, size_t lo, size_t hi, int stype, unsigned srel, uint32_t subj) {
    size_t end = hi;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        const tuple_t *t = &o->tup[mid];
        int lt = t->stype != stype ? t->stype < stype : (t->srel != srel ? t->srel < srel : t->subj < subj);
        if (lt) lo = mid + 1; else hi = mid;
    }
    if (lo < end && o->tup[lo].stype == stype && o->tup[lo].srel == srel && o->tup[lo].subj == subj) return lo;
    return end;
}

/* ---------------------------------------------------------------- evaluation
 * One C call of check_rel() == one SpiceDB dispatch (DispatchCheck).
 * depth_remaining follows dispatch.CheckDepth: a dispatch entered with 0
 * remaining fails with "max depth exceeded"; every nested dispatch (computed
 * userset, arrow target, userset subject) passes depth_remaining-1.
 * Result lattice (deterministic restatement of SpiceDB's racing set operations, see the
 * header): union HAS > ERR > NO; intersection NO > ERR > HAS; exclusion base first.
 */
typedef struct { int stype; unsigned srel; uint32_t sid; } subject_t;
enum { R_NO = 0, R_HAS = 1, R_ERR = 2 };

/* check_rel() is a pure function of (type, rel, id, depth_remaining) for a fixed
 * subject and snapshot, so its results may be memoised within one top-level
 * check without changing any answer.  SpiceDB itself does not memoise (caches
 * are off, pkg/spicedb/spicedb.go:45-47) and would, on cyclic data, burn its
 * whole dispatch budget; the memo only keeps THIS restatement polynomial.  It
 * switches on after MEMO_AFTER dispatches so ordinary shallow checks never pay
 * for it. */
#define MEMO_AFTER 4096
static uint64_t memo_key(int type, int rel, uint32_t id, int depth) {
    return ((uint64_t)(type & 0xFFF) << 52) | ((uint64_t)(rel & 0xFFF) << 40) | ((uint64_t)(depth & 0xFF) << 32) | id;
}
static void memo_reset(orc_t *o) {
    if (o->memo_n) memset(o->memo_k, 0xFF, sizeof(uint64_t) * o->memo_cap);
    o->memo_n = 0;
    o->memo_base = o->cnt_dispatch;
}
static int memo_get(orc_t *o, uint64_t key) {
    if (!o->memo_n) return -1;
    size_t h = (key * 0x9E3779B97F4A7C15ull) >> 17 & (o->memo_cap - 1);
    while (o->memo_k[h] != ~0ull) { if (o->memo_k[h] == key) return o->memo_v[h]; h = (h + 1) & (o->memo_cap - 1); }
    return -1;
}
static void memo_put(orc_t *o, uint64_t key, int val) {
    if ((o->memo_n + 1) * 2 > o->memo_cap) {
        size_t nc = o->memo_cap ? o->memo_cap * 2 : 4096;
        uint64_t *nk = malloc(sizeof(uint64_t) * nc);
        uint8_t *nv = malloc(nc);
        memset(nk, 0xFF, sizeof(uint64_t) * nc);
        for (size_t i = 0; i < o->memo_cap; i++)
            if (o->memo_k[i] != ~0ull) {
                size_t h = (o->memo_k[i] * 0x9E3779B97F4A7C15ull) >> 17 & (nc - 1);
                while (nk[h] != ~0ull) h = (h + 1) & (nc - 1);
                nk[h] = o->memo_k[i]; nv[h] = o->memo_v[i];
            }
        free(o->memo_k); free(o->memo_v);
        o->memo_k = nk; o->memo_v = nv; o->memo_cap = nc;
    }
    size_t h = (key * 0x9E3779B97F4A7C15ull) >> 17 & (o->memo_cap - 1);
    while (o->memo_k[h] != ~0ull) { if (o->memo_k[h] == key) { o->memo_v[h] = (uint8_t)val; return; } h = (h + 1) & (o->memo_cap - 1); }
    o->memo_k[h] = key; o->memo_v[h] = (uint8_t)val; o->memo_n++;
}

static int check_rel(orc_t *o, int type, int rel, uint32_t id, const subject_t *s, int depth_remaining);

static int join_union(int a, int b) {
    if (a == R_HAS || b == R_HAS) return R_HAS;
    if (a == R_ERR || b == R_ERR) return R_ERR;
    return R_NO;
}

static int eval_expr(orc_t *o, int type, const expr_t *e, uint32_t id, const subject_t *s, int depth_remaining) {
    type_t *t = &o->types[type];
    switch (e->kind) {
    case EX_NIL: return R_NO;
    case EX_INTERSECT: { /* `all`: the first empty operand decides; neither operand is a dispatch of its own */
        if (o->relaxed) { /* candidates: either operand's positive path reaches the subject */
            int ra = eval_expr(o, type, e->l, id, s, depth_remaining);
            if (ra == R_HAS) return R_HAS;
            return join_union(ra, eval_expr(o, type, e->r, id, s, depth_remaining));
        }
        int a = eval_expr(o, type, e->l, id, s, depth_remaining);
        if (a == R_NO) return R_NO;
        int b = eval_expr(o, type, e->r, id, s, depth_remaining);
        if (b == R_NO) return R_NO;
        return (a == R_ERR || b == R_ERR) ? R_ERR : R_HAS;
    }
    case EX_EXCLUDE: { /* `difference`: the base is waited for first; an empty base decides without the subtracted set */
        int a = eval_expr(o, type, e->l, id, s, depth_remaining);
        if (a != R_HAS || o->relaxed) return a; /* (candidates: the subtracted operand is not a positive occurrence) */
        int b = eval_expr(o, type, e->r, id, s, depth_remaining);
        return b == R_ERR ? R_ERR : (b == R_HAS ? R_NO : R_HAS);
    }
    case EX_UNION: {
        int a = eval_expr(o, type, e->l, id, s, depth_remaining);
        if (a == R_HAS) return R_HAS;
        return join_union(a, eval_expr(o, type, e->r, id, s, depth_remaining));
    }
    case EX_REF: /* computed userset: dispatch on the same object */
        return check_rel(o, type, rel_index(t, e->a), id, s, depth_remaining - 1);
    case EX_ARROW: { /* tuple-to-userset */
        int ts = rel_index(t, e->a);
        size_t lo, hi;
        row_range(o, type, ts, id, &lo, &hi);
        o->cnt_rows++;
        int acc = R_NO;
        for (size_t i = lo; i < hi; i++) {
            const tuple_t *tp = &o->tup[i];
            if (!tup_live(o, tp)) continue;
            o->cnt_edges++;
            int tr = rel_index(&o->types[tp->stype], e->b);
            if (tr < 0) continue; /* subject type lacks the computed relation: skipped */
            int r = check_rel(o, tp->stype, tr, tp->subj, s, depth_remaining - 1);
            if (r == R_HAS) return R_HAS;
            acc = join_union(acc, r);
        }
        return acc;
    }
    case EX_ARROW_ALL: { /* intersection arrow (EXTERNAL, unverified: restated from SpiceDB's documentation of `.all()`): every subject of the
                          * tupleset is dispatched; no subject -> NO; one NO decides; else an error wins over HAS.  A subject whose type lacks
                          * the computed permission is not dispatched, as for `->` */
        int ts = rel_index(t, e->a);
        size_t lo, hi;
        row_range(o, type, ts, id, &lo, &hi);
        o->cnt_rows++;
        int n = 0, any_err = 0;
        for (size_t i = lo; i < hi; i++) {
            const tuple_t *tp = &o->tup[i];
            if (!tup_live(o, tp)) continue;
            o->cnt_edges++;
            int tr = rel_index(&o->types[tp->stype], e->b);
            if (tr < 0) continue;
            int r = check_rel(o, tp->stype, tr, tp->subj, s, depth_remaining - 1);
            if (o->relaxed) { /* candidates: as `->` */
                if (r == R_HAS) return R_HAS;
                if (r == R_ERR) any_err = 1;
                continue;
            }
            if (r == R_NO) return R_NO;
            if (r == R_ERR) any_err = 1;
            n++;
        }
        if (o->relaxed) return any_err ? R_ERR : R_NO;
        return n == 0 ? R_NO : (any_err ? R_ERR : R_HAS);
    }
    }
    return R_NO;
}

static int check_rel_body(orc_t *o, int type, int rel, uint32_t id, const subject_t *s, int depth_remaining);
static int check_rel(orc_t *o, int type, int rel, uint32_t id, const subject_t *s, int depth_remaining) {
    if (depth_remaining <= 0) return R_ERR; /* max depth exceeded */
    if (o->cnt_dispatch - o->memo_base < MEMO_AFTER) return check_rel_body(o, type, rel, id, s, depth_remaining);
    uint64_t key = memo_key(type, rel, id, depth_remaining);
    int m = memo_get(o, key);
    if (m >= 0) return m;
    m = check_rel_body(o, type, rel, id, s, depth_remaining);
    memo_put(o, key, m);
    return m;
}
static int check_rel_body(orc_t *o, int type, int rel, uint32_t id, const subject_t *s, int depth_remaining) {
    o->cnt_dispatch++;
    /* the subject itself, when it is exactly this object#relation, is a member */
    if (s->stype == type && s->srel == (unsigned)rel && s->sid == id) return R_HAS;
    rel_t *r = &o->types[type].rels[rel];
    if (r->is_perm) return eval_expr(o, type, r->expr, id, s, depth_remaining);
    /* checkDirect */
    size_t lo, hi;
    row_range(o, type, rel, id, &lo, &hi);
    o->cnt_rows++;
    int acc = R_NO;
    { /* direct (terminal) subject match */
        size_t p = row_find(o, lo, hi, s->stype, s->srel, s->sid);
        o->cnt_edges += hi - lo;
        if (p < hi && tup_live(o, &o->tup[p])) return R_HAS;
        if (s->srel == ELLIPSIS) { /* `T:*` on the row: every plain subject of type T is a member */
            for (size_t i = lo; i < hi; i++)
                if (o->tup[i].stype == s->stype && o->tup[i].srel == WILDCARD && tup_live(o, &o->tup[i])) return R_HAS;
        }
    }
    for (size_t i = lo; i < hi; i++) { /* userset subjects: dispatch */
        const tuple_t *tp = &o->tup[i];
        if (!tup_live(o, tp) || tp->srel == ELLIPSIS || tp->srel == WILDCARD) continue;
        int x = check_rel(o, tp->stype, (int)tp->srel, tp->subj, s, depth_remaining - 1);
        if (x == R_HAS) return R_HAS;
        acc = join_union(acc, x);
    }
    return acc;
}

/* --------------------------------------------------------------- public API */
orc_t *orc_new(const char *schema_text, char *errbuf, int errlen) {
    orc_t *o = calloc(1, sizeof *o);
    o->sorted = 1;
    o->revision = 1;
    if (!parse_schema(o, schema_text)) {
        if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "%s", o->err);
        /* leak-free teardown */
        extern void orc_free(orc_t *);
        orc_free(o);
        return NULL;
    }
    return o;
}

void orc_free(orc_t *o) {
    if (!o) return;
    for (int i = 0; i < o->ntypes; i++) {
        type_t *t = &o->types[i];
        for (int j = 0; j < t->nrels; j++) {
            rel_t *r = &t->rels[j];
            for (int k = 0; k < r->nallowed; k++) { free(r->a_type[k]); free(r->a_rel[k]); }
            free(r->a_type); free(r->a_rel); free(r->a_exp); free(r->a_wild); free(r->allowed);
            ex_free(r->expr);
            free(r->name);
        }
        free(t->rels);
        st_free(&t->objs);
        free(t->name);
    }
    free(o->types);
    free(o->tup);
    free(o->lr_ids);
    free(o->memo_k);
    free(o->memo_v);
    tn_drop(o);
    free(o);
}

const char *orc_last_error(orc_t *o) { return o->err; }
void orc_set_now(orc_t *o, int64_t unix_seconds) { o->now = unix_seconds; }
uint64_t orc_revision(orc_t *o) { return o->revision; }
size_t orc_num_tuples(orc_t *o) { ensure_sorted(o); return o->ntup; }
int orc_type_id(orc_t *o, const char *type) { return type_index(o, type); }
int orc_rel_id(orc_t *o, int type, const char *rel) {
    if (type < 0 || type >= o->ntypes) return -1;
    return rel_index(&o->types[type], rel);
}
uint32_t orc_intern(orc_t *o, int type, const char *id) { return st_intern(&o->types[type].objs, id); }
const char *orc_object_name(orc_t *o, int type, uint32_t id) {
    if (type < 0 || type >= o->ntypes || id >= o->types[type].objs.n) return NULL;
    return o->types[type].objs.strs[id];
}
void orc_counters(orc_t *o, uint64_t *dispatches, uint64_t *rows, uint64_t *edges) {
    *dispatches = o->cnt_dispatch; *rows = o->cnt_rows; *edges = o->cnt_edges;
}

/* is (stype, srel) an allowed subject of relation (rtype, rel)?  returns allowed_t* or NULL */
static const allowed_t *allowed_subject(orc_t *o, int rtype, int rel, int stype, unsigned srel) {
    rel_t *r = &o->types[rtype].rels[rel];
    for (int k = 0; k < r->nallowed; k++)
        if (r->allowed[k].stype == stype && r->allowed[k].srel == srel) return &r->allowed[k];
    return NULL;
}

/* Bulk numeric load: object ids are caller-chosen dense integers (no strings).
 * Appends; call orc_freeze() (or any read) afterwards. */
int orc_add_edges(orc_t *o, int rtype, int rel, int stype, int srel, size_t n, const uint32_t *res, const uint32_t *subj) {
    unsigned sr = srel == -2 ? WILDCARD : (srel < 0 ? ELLIPSIS : (unsigned)srel); /* -2: `stype:*` relationships (subj[] is ignored) */
    if (rtype < 0 || rtype >= o->ntypes || rel < 0 || rel >= o->types[rtype].nrels || o->types[rtype].rels[rel].is_perm ||
        !allowed_subject(o, rtype, rel, stype, sr)) {
        seterr(o, "add_edges: subject type not allowed on relation");
        return ORC_ERR_FAILED_PRECONDITION;
    }
    if (o->ntup + n > o->captup) {
        o->captup = (o->ntup + n) * 3 / 2 + 16;
        o->tup = realloc(o->tup, sizeof(tuple_t) * o->captup);
    }
    for (size_t i = 0; i < n; i++) {
        tuple_t *t = &o->tup[o->ntup++];
        t->rtype = (uint16_t)rtype; t->rel = (uint16_t)rel; t->res = res[i];
        t->stype = (uint16_t)stype; t->srel = (uint16_t)sr; t->subj = sr == WILDCARD ? 0u : subj[i];
        t->expires = 0;
    }
    o->sorted = 0;
    o->revision++;
    return ORC_OK;
}
void orc_freeze(orc_t *o) { ensure_sorted(o); }

/* ---- text-level relationship structs (mirror authzed.api.v1 messages) ---- */
typedef struct {
    const char *rtype, *rid, *rel, *stype, *sid, *srel; /* srel NULL or "" = none */
    int64_t expires_at;                                   /* unix seconds, 0 = none */
} orc_rel_t;
typedef struct { int op; orc_rel_t rel; } orc_update_t;
typedef struct {
    int op;              /* ORC_PRE_* */
    const char *rtype;   /* required */
    const char *rid;     /* NULL/"" = any */
    const char *rel;     /* NULL/"" = any */
    const char *stype;   /* NULL = no subject filter */
    const char *sid;     /* NULL/"" = any */
    const char *srel;    /* NULL = any relation; "" = only ellipsis; else exact */
} orc_filter_t;

static int resolve_rel(orc_t *o, const orc_rel_t *r, tuple_t *out, int intern) {
    if (!r->rtype || !r->rid || !r->rel || !r->stype || !r->sid || !*r->rtype || !*r->rid || !*r->rel || !*r->stype || !*r->sid) {
        seterr(o, "invalid relationship: empty field");
        return ORC_ERR_INVALID_ARGUMENT;
    }
    if (!names_wellformed(o, r->rtype, r->rel, r->stype, r->srel) || !ok_id(r->rid) || (strcmp(r->sid, "*") != 0 && !ok_id(r->sid))) {
        seterr(o, "invalid relationship: a field does not match the API's pattern");
        return ORC_ERR_INVALID_ARGUMENT;
    }
    int rt = type_index(o, r->rtype);
    if (rt < 0) { seterr(o, "object definition `%s` not found", r->rtype); return ORC_ERR_FAILED_PRECONDITION; }
    int rl = rel_index(&o->types[rt], r->rel);
    if (rl < 0) { seterr(o, "relation/permission `%s` not found under definition `%s`", r->rel, r->rtype); return ORC_ERR_FAILED_PRECONDITION; }
    int st = type_index(o, r->stype);
    if (st < 0) { seterr(o, "object definition `%s` not found", r->stype); return ORC_ERR_FAILED_PRECONDITION; }
    unsigned sr = ELLIPSIS;
    if (r->srel && *r->srel && strcmp(r->srel, "...") != 0) {
        int x = rel_index(&o->types[st], r->srel);
        if (x < 0) { seterr(o, "relation/permission `%s` not found under definition `%s`", r->srel, r->stype); return ORC_ERR_FAILED_PRECONDITION; }
        sr = (unsigned)x;
    }
    if (strcmp(r->sid, "*") == 0) { /* `T:*`: a class of its own; never with a subject relation */
        if (sr != ELLIPSIS) { seterr(o, "invalid relationship: wildcard subject with a relation"); return ORC_ERR_INVALID_ARGUMENT; }
        sr = WILDCARD;
    }
    out->rtype = (uint16_t)rt; out->rel = (uint16_t)rl; out->stype = (uint16_t)st; out->srel = (uint16_t)sr;
    if (intern) {
        out->res = st_intern(&o->types[rt].objs, r->rid);
        out->subj = sr == WILDCARD ? 0u : st_intern(&o->types[st].objs, r->sid);
    } else {
        out->res = st_find(&o->types[rt].objs, r->rid);
        out->subj = sr == WILDCARD ? 0u : st_find(&o->types[st].objs, r->sid);
    }
    out->expires = r->expires_at;
    return ORC_OK;
}

static int filter_match(orc_t *o, const orc_filter_t *f, const tuple_t *t) {
    if (!tup_live(o, t)) return 0;
    if (strcmp(o->types[t->rtype].name, f->rtype) != 0) return 0;
    if (f->rid && *f->rid && strcmp(o->types[t->rtype].objs.strs[t->res], f->rid) != 0) return 0;
    if (f->rel && *f->rel && strcmp(o->types[t->rtype].rels[t->rel].name, f->rel) != 0) return 0;
    if (f->stype) {
        if (strcmp(o->types[t->stype].name, f->stype) != 0) return 0;
        if (f->sid && *f->sid && strcmp(t->srel == WILDCARD ? "*" : o->types[t->stype].objs.strs[t->subj], f->sid) != 0) return 0;
        if (f->srel) {
            if (!*f->srel || strcmp(f->srel, "...") == 0) { if (t->srel != ELLIPSIS && t->srel != WILDCARD) return 0; }
            else {
                if (t->srel == ELLIPSIS || t->srel == WILDCARD) return 0;
                if (strcmp(o->types[t->stype].rels[t->srel].name, f->srel) != 0) return 0;
            }
        }
    }
    return 1;
}

static int filter_valid(orc_t *o, const orc_filter_t *f) {
    if (!f->rtype || !*f->rtype) { seterr(o, "filter: resource type required"); return ORC_ERR_INVALID_ARGUMENT; }
    if (!names_wellformed(o, f->rtype, f->rel, f->stype, f->stype ? f->srel : NULL) || (f->rid && *f->rid && !ok_id(f->rid)) ||
        (f->stype && f->sid && *f->sid && strcmp(f->sid, "*") != 0 && !ok_id(f->sid))) {
        seterr(o, "filter: a field does not match the API's pattern");
        return ORC_ERR_INVALID_ARGUMENT;
    }
    int rt = type_index(o, f->rtype);
    if (rt < 0) { seterr(o, "object definition `%s` not found", f->rtype); return ORC_ERR_FAILED_PRECONDITION; }
    if (f->rel && *f->rel && rel_index(&o->types[rt], f->rel) < 0) {
        seterr(o, "relation `%s` not found under `%s`", f->rel, f->rtype);
        return ORC_ERR_FAILED_PRECONDITION;
    }
    if (f->stype && type_index(o, f->stype) < 0) { seterr(o, "object definition `%s` not found", f->stype); return ORC_ERR_FAILED_PRECONDITION; }
    return ORC_OK;
}

/* WriteRelationships: all preconditions are evaluated against the pre-write
 * state; the whole request is atomic (reference: activity.go:47-77,
 * workflow.go:452-462; limits spicedb.go:35-36). */
int orc_write(orc_t *o, const orc_update_t *ups, int nups, const orc_filter_t *pre, int npre, uint64_t *revision_out) {
    ensure_sorted(o);
    if (nups > 1000) { seterr(o, "update count of %d is greater than maximum allowed of 1000", nups); return ORC_ERR_INVALID_ARGUMENT; }
    if (npre > 1000) { seterr(o, "precondition count of %d is greater than maximum allowed of 1000", npre); return ORC_ERR_INVALID_ARGUMENT; }
    for (int i = 0; i < npre; i++) {
        int rc = filter_valid(o, &pre[i]);
        if (rc) return rc;
    }
    tuple_t *res = malloc(sizeof(tuple_t) * (nups ? nups : 1));
    for (int i = 0; i < nups; i++) {
        if (ups[i].op < ORC_OP_CREATE || ups[i].op > ORC_OP_DELETE) { free(res); seterr(o, "invalid update operation"); return ORC_ERR_INVALID_ARGUMENT; }
        int rc = resolve_rel(o, &ups[i].rel, &res[i], 1);
        if (rc) { free(res); return rc; }
        tuple_t *t = &res[i];
        if (o->types[t->rtype].rels[t->rel].is_perm) {
            free(res); seterr(o, "cannot write a relationship to permission `%s`", ups[i].rel.rel);
            return ORC_ERR_INVALID_ARGUMENT;
        }
        const allowed_t *a = allowed_subject(o, t->rtype, t->rel, t->stype, t->srel);
        if (!a) {
            free(res);
            seterr(o, "subjects of type `%s%s%s` are not allowed on relation `%s#%s`", ups[i].rel.stype,
                   t->srel == WILDCARD ? ":*" : (t->srel == ELLIPSIS ? "" : "#"), t->srel >= WILDCARD ? "" : ups[i].rel.srel, ups[i].rel.rtype, ups[i].rel.rel);
            return ORC_ERR_INVALID_ARGUMENT;
        }
        if (t->expires && !a->expiring) { free(res); seterr(o, "relation does not allow expiration"); return ORC_ERR_INVALID_ARGUMENT; }
        for (int j = 0; j < i; j++)
            if (tup_cmp(&res[j], t) == 0) { free(res); seterr(o, "found duplicate update for the same relationship"); return ORC_ERR_INVALID_ARGUMENT; }
    }
    for (int i = 0; i < npre; i++) {
        int any = 0;
        for (size_t k = 0; k < o->ntup && !any; k++) any = filter_match(o, &pre[i], &o->tup[k]);
        if ((pre[i].op == ORC_PRE_MUST_MATCH && !any) || (pre[i].op == ORC_PRE_MUST_NOT_MATCH && any)) {
            free(res); seterr(o, "unable to satisfy write precondition"); return ORC_ERR_FAILED_PRECONDITION;
        }
    }
    for (int i = 0; i < nups; i++) {
        if (ups[i].op != ORC_OP_CREATE) continue;
        size_t p = lower_bound(o, &res[i]);
        if (p < o->ntup && tup_cmp(&o->tup[p], &res[i]) == 0 && tup_live(o, &o->tup[p])) {
            free(res); seterr(o, "could not CREATE relationship, as it already existed"); return ORC_ERR_ALREADY_EXISTS;
        }
    }
    for (int i = 0; i < nups; i++) {
        size_t p = lower_bound(o, &res[i]);
        int present = p < o->ntup && tup_cmp(&o->tup[p], &res[i]) == 0;
        if (ups[i].op == ORC_OP_DELETE) {
            if (present) { memmove(&o->tup[p], &o->tup[p + 1], sizeof(tuple_t) * (o->ntup - p - 1)); o->ntup--; }
        } else if (present) {
            o->tup[p] = res[i];
        } else {
            if (o->ntup == o->captup) { o->captup = o->captup ? o->captup * 2 : 64; o->tup = realloc(o->tup, sizeof(tuple_t) * o->captup); }
            memmove(&o->tup[p + 1], &o->tup[p], sizeof(tuple_t) * (o->ntup - p));
            o->tup[p] = res[i];
            o->ntup++;
        }
    }
    free(res);
    o->revision++;
    if (revision_out) *revision_out = o->revision;
    return ORC_OK;
}

/* DeleteRelationships by filter (e2e/util_test.go:66); returns count deleted via *ndeleted */
int orc_delete_by_filter(orc_t *o, const orc_filter_t *f, uint64_t *ndeleted) {
    ensure_sorted(o);
    int rc = filter_valid(o, f);
    if (rc) return rc;
    size_t w = 0, del = 0;
    for (size_t i = 0; i < o->ntup; i++) {
        if (filter_match(o, f, &o->tup[i])) { del++; continue; }
        o->tup[w++] = o->tup[i];
    }
    o->ntup = w;
    o->revision++;
    if (ndeleted) *ndeleted = del;
    return ORC_OK;
}

/* ReadRelationships: calls cb(user, "type:id#rel@stype:sid[#srel]", expires) per match */
typedef void (*orc_read_cb)(void *user, const char *rtype, const char *rid, const char *rel, const char *stype, const char *sid,
                            const char *srel, int64_t expires_at);
int orc_read(orc_t *o, const orc_filter_t *f, orc_read_cb cb, void *user) {
    ensure_sorted(o);
    int rc = filter_valid(o, f);
    if (rc) return rc;
    for (size_t i = 0; i < o->ntup; i++) {
        const tuple_t *t = &o->tup[i];
        if (!filter_match(o, f, t)) continue;
        cb(user, o->types[t->rtype].name, o->types[t->rtype].objs.strs[t->res], o->types[t->rtype].rels[t->rel].name,
           o->types[t->stype].name, t->srel == WILDCARD ? "*" : o->types[t->stype].objs.strs[t->subj],
           t->srel >= WILDCARD ? "" : o->types[t->stype].rels[t->srel].name, t->expires);
    }
    return ORC_OK;
}

/* CheckPermission on strings.  Returns permissionship (ORC_PERM_*); *err = 0 or ORC_ERR_*. */
int orc_check(orc_t *o, const char *rtype, const char *rid, const char *perm, const char *stype, const char *sid, const char *srel, int *err) {
    ensure_sorted(o);
    *err = 0;
    o->cnt_dispatch = o->cnt_rows = o->cnt_edges = 0;
    if (!rtype || !rid || !perm || !stype || !sid || !*rtype || !*rid || !*perm || !*stype || !*sid) {
        seterr(o, "invalid CheckPermissionRequest: empty field");
        *err = ORC_ERR_INVALID_ARGUMENT; /* KAT-3: pkg/proxy/options_test.go:101-102 */
        return ORC_PERM_UNSPEC;
    }
    if (!names_wellformed(o, rtype, perm, stype, srel) || !ok_id(rid) || !ok_id(sid)) { /* (`*` is not an object id in a Check) */
        seterr(o, "invalid CheckPermissionRequest: a field does not match the API's pattern");
        *err = ORC_ERR_INVALID_ARGUMENT;
        return ORC_PERM_UNSPEC;
    }
    int rt = type_index(o, rtype);
    if (rt < 0) { seterr(o, "object definition `%s` not found", rtype); *err = ORC_ERR_FAILED_PRECONDITION; return ORC_PERM_UNSPEC; }
    int rl = rel_index(&o->types[rt], perm);
    if (rl < 0) { seterr(o, "relation/permission `%s` not found under definition `%s`", perm, rtype); *err = ORC_ERR_FAILED_PRECONDITION; return ORC_PERM_UNSPEC; }
    int st = type_index(o, stype);
    if (st < 0) { seterr(o, "object definition `%s` not found", stype); *err = ORC_ERR_FAILED_PRECONDITION; return ORC_PERM_UNSPEC; }
    unsigned sr = ELLIPSIS;
    if (srel && *srel && strcmp(srel, "...") != 0) {
        int x = rel_index(&o->types[st], srel);
        if (x < 0) { seterr(o, "relation/permission `%s` not found under definition `%s`", srel, stype); *err = ORC_ERR_FAILED_PRECONDITION; return ORC_PERM_UNSPEC; }
        sr = (unsigned)x;
    }
    /* unknown object ids simply have no relationships.  Use sentinel ids that
     * can never collide with interned ones (distinct for resource / subject). */
    uint32_t res = st_find(&o->types[rt].objs, rid);
    uint32_t sub = st_find(&o->types[st].objs, sid);
    if (res == 0xFFFFFFFFu && sub == 0xFFFFFFFFu && rt == st && strcmp(rid, sid) == 0) { res = sub = 0xFFFFFFFEu; }
    else { if (res == 0xFFFFFFFFu) res = 0xFFFFFFFDu; if (sub == 0xFFFFFFFFu) sub = 0xFFFFFFFCu; }
    subject_t s = {st, sr, sub};
    memo_reset(o);
    int r = check_rel(o, rt, rl, res, &s, ORC_MAX_DEPTH);
    if (r == R_ERR) { seterr(o, "max depth exceeded: this usually indicates a recursive or too deep data dependency"); *err = ORC_ERR_DEPTH; return ORC_PERM_UNSPEC; }
    return r == R_HAS ? ORC_PERM_HAS : ORC_PERM_NO;
}

/* Numeric bulk check (ids as given to orc_add_edges).  out[i] = ORC_PERM_*, err[i] = 0 / ORC_ERR_DEPTH. */
void orc_check_bulk_ids(orc_t *o, size_t n, int rtype, int perm, const uint32_t *res, int stype, int srel, const uint32_t *subj,
                        uint8_t *out, int32_t *err) {
    ensure_sorted(o);
    o->cnt_dispatch = o->cnt_rows = o->cnt_edges = 0;
    for (size_t i = 0; i < n; i++) {
        subject_t s = {stype, srel < 0 ? ELLIPSIS : (unsigned)srel, subj[i]};
        memo_reset(o);
        int r = check_rel(o, rtype, perm, res[i], &s, ORC_MAX_DEPTH);
        out[i] = r == R_HAS ? ORC_PERM_HAS : (r == R_ERR ? ORC_PERM_UNSPEC : ORC_PERM_NO);
        if (err) err[i] = r == R_ERR ? ORC_ERR_DEPTH : 0;
    }
}

/* Same, split statically over `nthreads` host threads (the "all host cores" CPU baseline of SURVEY.md 8(d)).
 * Every thread evaluates with a private shallow copy of the handle: the frozen tuple and type tables are only
 * read, the per-call memo and the work counters are per thread. */
typedef struct {
    orc_t ctx;
    size_t lo, hi;
    int rtype, perm, stype, srel;
    const uint32_t *res, *subj;
    uint8_t *out;
    int32_t *err;
} mt_job_t;
static void *mt_run(void *p) {
    mt_job_t *j = (mt_job_t *)p;
    for (size_t i = j->lo; i < j->hi; i++) {
        subject_t s = {j->stype, j->srel < 0 ? ELLIPSIS : (unsigned)j->srel, j->subj[i]};
        memo_reset(&j->ctx);
        int r = check_rel(&j->ctx, j->rtype, j->perm, j->res[i], &s, ORC_MAX_DEPTH);
        j->out[i] = r == R_HAS ? ORC_PERM_HAS : (r == R_ERR ? ORC_PERM_UNSPEC : ORC_PERM_NO);
        if (j->err) j->err[i] = r == R_ERR ? ORC_ERR_DEPTH : 0;
    }
    return NULL;
}
void orc_check_bulk_ids_mt(orc_t *o, int nthreads, size_t n, int rtype, int perm, const uint32_t *res, int stype, int srel, const uint32_t *subj,
                           uint8_t *out, int32_t *err) {
    ensure_sorted(o);
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n && n) nthreads = (int)n;
    mt_job_t *jobs = calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *th = calloc((size_t)nthreads, sizeof *th);
    for (int t = 0; t < nthreads; t++) {
        mt_job_t *j = &jobs[t];
        j->ctx = *o;
        j->ctx.memo_k = NULL; j->ctx.memo_v = NULL; j->ctx.memo_cap = j->ctx.memo_n = 0;
        j->ctx.lr_ids = NULL; j->ctx.lr_n = j->ctx.lr_cap = 0;
        j->ctx.cnt_dispatch = j->ctx.cnt_rows = j->ctx.cnt_edges = 0;
        j->lo = n * (size_t)t / (size_t)nthreads;
        j->hi = n * (size_t)(t + 1) / (size_t)nthreads;
        j->rtype = rtype; j->perm = perm; j->stype = stype; j->srel = srel;
        j->res = res; j->subj = subj; j->out = out; j->err = err;
        pthread_create(&th[t], NULL, mt_run, j);
    }
    o->cnt_dispatch = o->cnt_rows = o->cnt_edges = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        o->cnt_dispatch += jobs[t].ctx.cnt_dispatch; o->cnt_rows += jobs[t].ctx.cnt_rows; o->cnt_edges += jobs[t].ctx.cnt_edges;
        free(jobs[t].ctx.memo_k); free(jobs[t].ctx.memo_v);
    }
    free(jobs); free(th);
}

/* LookupResources restated as its definition: { id : Check(T:id#p @ S) == HAS }
 * evaluated by brute force over every object of type T that occurs as a
 * resource in any tuple (plus the subject itself for the reflexive case).
 * Returns the number of ids; fetch with orc_lookup_result().
 *
 * Errors (reference pkg/authz/lookups.go:75-83: the stream ends at the first Recv error and the list request fails with it,
 * responsefilterer.go:196-204).  EXTERNAL, unverified: SpiceDB finds CANDIDATES by reverse reachability from the subject and confirms the
 * ones that passed an intersection / exclusion with a Check; a Check that errs (a branch beyond the dispatch depth under `&` / `-`) ends the
 * stream.  Restated: a resource whose Check is ERR fails the whole lookup iff it is a candidate -- iff the positive relaxation of the
 * schema (o->relaxed) grants it.  Resources that merely sit on a cycle or a long chain the subject has nothing to do with are not
 * candidates and stay out of the answer silently.  Returns -1 (orc_lookup_error() = the code) on failure. */
static int u32_cmp(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
static void lr_push(orc_t *o, uint32_t id) {
    if (o->lr_n == o->lr_cap) { o->lr_cap = o->lr_cap ? o->lr_cap * 2 : 256; o->lr_ids = realloc(o->lr_ids, sizeof(uint32_t) * o->lr_cap); }
    o->lr_ids[o->lr_n++] = id;
}
long orc_lookup_ids(orc_t *o, int rtype, int perm, int stype, int srel, uint32_t subj) {
    ensure_sorted(o);
    o->lr_n = 0;
    subject_t s = {stype, srel < 0 ? ELLIPSIS : (unsigned)srel, subj};
    /* candidate ids: distinct resource ids of type rtype */
    size_t ncand = 0, capc = 1024;
    uint32_t *cand = malloc(sizeof(uint32_t) * capc);
    for (size_t i = 0; i < o->ntup; i++) {
        if (o->tup[i].rtype != rtype) continue;
        if (ncand == capc) { capc *= 2; cand = realloc(cand, sizeof(uint32_t) * capc); }
        cand[ncand++] = o->tup[i].res;
    }
    if (stype == rtype) { if (ncand == capc) { capc *= 2; cand = realloc(cand, sizeof(uint32_t) * capc); } cand[ncand++] = subj; }
    qsort(cand, ncand, sizeof(uint32_t), u32_cmp);
    for (size_t i = 0; i < ncand; i++) {
        if (i && cand[i] == cand[i - 1]) continue;
        memo_reset(o);
        int r = check_rel(o, rtype, perm, cand[i], &s, ORC_MAX_DEPTH);
        if (r == R_HAS) lr_push(o, cand[i]);
        else if (r == R_ERR && !o->lenient_lookup) {
            memo_reset(o);
            o->relaxed = 1;
            int rr = check_rel(o, rtype, perm, cand[i], &s, ORC_MAX_DEPTH);
            o->relaxed = 0;
            memo_reset(o);
            if (rr == R_HAS) {
                seterr(o, "LookupResources: max depth exceeded while checking candidate id %u", cand[i]);
                o->lookup_err = ORC_ERR_DEPTH;
                free(cand);
                o->lr_n = 0;
                return -1;
            }
        }
    }
    free(cand);
    return (long)o->lr_n;
}
int orc_lookup_error(orc_t *o) { return o->lookup_err; }
void orc_set_lenient_lookup(orc_t *o, int on) { o->lenient_lookup = on; }
const uint32_t *orc_lookup_result(orc_t *o) { return o->lr_ids; }

long orc_lookup(orc_t *o, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, int *err) {
    *err = 0;
    o->lr_n = 0;
    if (!rtype || !perm || !stype || !sid || !*rtype || !*perm || !*stype || !*sid) { seterr(o, "invalid LookupResourcesRequest"); *err = ORC_ERR_INVALID_ARGUMENT; return -1; }
    if (!names_wellformed(o, rtype, perm, stype, srel) || !ok_id(sid)) { seterr(o, "invalid LookupResourcesRequest: a field does not match the API's pattern"); *err = ORC_ERR_INVALID_ARGUMENT; return -1; }
    int rt = type_index(o, rtype);
    if (rt < 0) { seterr(o, "object definition `%s` not found", rtype); *err = ORC_ERR_FAILED_PRECONDITION; return -1; }
    int rl = rel_index(&o->types[rt], perm);
    if (rl < 0) { seterr(o, "relation/permission `%s` not found", perm); *err = ORC_ERR_FAILED_PRECONDITION; return -1; }
    int st = type_index(o, stype);
    if (st < 0) { seterr(o, "object definition `%s` not found", stype); *err = ORC_ERR_FAILED_PRECONDITION; return -1; }
    int sr = -1;
    if (srel && *srel && strcmp(srel, "...") != 0) {
        sr = rel_index(&o->types[st], srel);
        if (sr < 0) { seterr(o, "relation `%s` not found", srel); *err = ORC_ERR_FAILED_PRECONDITION; return -1; }
    }
    /* the subject may be unknown to the store: intern it so the reflexive case has an id */
    uint32_t sub = st_intern(&o->types[st].objs, sid);
    long n = orc_lookup_ids(o, rt, rl, st, sr, sub);
    if (n < 0) *err = o->lookup_err;
    return n;
}

/* ------------------------------------------------ algorithmic-bytes model
 * bytes a level-synchronous evaluation must move for one
 * Check:  17 + sum_{rows touched} 8 + sum_{enumerated rows} 4*deg
 *            + sum_{membership rows} 4*ceil(log2(deg+1)),
 * rows = distinct (edge class, resource) CSR rows, evaluation stops after the
 * level in which HAS is established.  Levels are dispatch levels.  Returns the
 * byte count; *result = permissionship. */
typedef struct { uint64_t *k; size_t cap, n; } u64set_t;
static int set_add(u64set_t *s, uint64_t key) { /* returns 1 if newly added */
    if ((s->n + 1) * 2 > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 256;
        uint64_t *nk = malloc(sizeof(uint64_t) * nc);
        memset(nk, 0xFF, sizeof(uint64_t) * nc);
        for (size_t i = 0; i < s->cap; i++)
            if (s->k[i] != ~0ull) { size_t h = (s->k[i] * 0x9E3779B97F4A7C15ull) >> 20 & (nc - 1); while (nk[h] != ~0ull) h = (h + 1) & (nc - 1); nk[h] = s->k[i]; }
        free(s->k); s->k = nk; s->cap = nc;
    }
    size_t h = (key * 0x9E3779B97F4A7C15ull) >> 20 & (s->cap - 1);
    while (s->k[h] != ~0ull) { if (s->k[h] == key) return 0; h = (h + 1) & (s->cap - 1); }
    s->k[h] = key; s->n++;
    return 1;
}
static unsigned clog2(uint64_t x) { unsigned b = 0; while ((1ull << b) < x) b++; return b; } /* ceil(log2(x)), x>=1 */

typedef struct { uint16_t type, rel; uint32_t id; } state_t;

/* lvl_bytes[L] / lvl_states[L] (L = 1..ORC_MAX_DEPTH, may be NULL): bytes charged while level L was evaluated and the
 * number of distinct states that level held (level 1 = the request itself; its 17 request/answer bytes are charged there). */
static uint64_t check_bytes_core(orc_t *o, int rtype, int perm, uint32_t res, int stype, int srel, uint32_t subj, int *result, uint64_t *lvl_bytes,
                                 uint64_t *lvl_states) {
    subject_t s = {stype, srel < 0 ? ELLIPSIS : (unsigned)srel, subj};
    uint64_t bytes = 17;
    u64set_t seen_states = {0}, seen_rows = {0};
    size_t fcap = 256, fn = 0, nn = 0;
    state_t *front = malloc(sizeof(state_t) * fcap), *next = malloc(sizeof(state_t) * fcap);
    front[fn++] = (state_t){(uint16_t)rtype, (uint16_t)perm, res};
    set_add(&seen_states, ((uint64_t)rtype << 48) | ((uint64_t)perm << 32) | res);
    int found = 0;
#define PUSH_NEXT(T, R, I)                                                                                          \
    do {                                                                                                            \
        if (set_add(&seen_states, ((uint64_t)(T) << 48) | ((uint64_t)(R) << 32) | (I))) {                           \
            if (nn == fcap) { fcap *= 2; front = realloc(front, sizeof(state_t) * fcap); next = realloc(next, sizeof(state_t) * fcap); } \
            next[nn++] = (state_t){(uint16_t)(T), (uint16_t)(R), (I)};                                              \
        }                                                                                                           \
    } while (0)
    uint64_t charged = 0;
    for (int level = 1; level <= ORC_MAX_DEPTH && fn && !found; level++) {
        nn = 0;
        if (lvl_states) lvl_states[level] += fn;
        for (size_t fi = 0; fi < fn; fi++) {
            state_t st = front[fi];
            if (s.stype == st.type && s.srel == st.rel && s.sid == st.id) found = 1;
            rel_t *r = &o->types[st.type].rels[st.rel];
            if (!r->is_perm) {
                size_t lo, hi;
                row_range(o, st.type, st.rel, st.id, &lo, &hi);
                /* one CSR row per allowed subject class of this relation */
                for (int k = 0; k < r->nallowed; k++) {
                    const allowed_t *a = &r->allowed[k];
                    uint64_t rowkey = ((uint64_t)st.type << 56) ^ ((uint64_t)st.rel << 48) ^ ((uint64_t)k << 40) ^ ((uint64_t)st.id);
                    rowkey ^= 0x8000000000000000ull;
                    uint64_t deg = 0;
                    for (size_t i = lo; i < hi; i++)
                        if (o->tup[i].stype == a->stype && o->tup[i].srel == a->srel && tup_live(o, &o->tup[i])) deg++;
                    int fresh = set_add(&seen_rows, rowkey);
                    if (fresh) bytes += 8;
                    int is_member_class = a->stype == s.stype && a->srel == s.srel;
                    if (a->srel < WILDCARD) { /* userset class: enumerate */
                        if (fresh) bytes += 4 * deg;
                        for (size_t i = lo; i < hi; i++) {
                            const tuple_t *tp = &o->tup[i];
                            if (tp->stype != a->stype || tp->srel != a->srel || !tup_live(o, tp)) continue;
                            if (is_member_class && tp->subj == s.sid) found = 1;
                            PUSH_NEXT(tp->stype, tp->srel, tp->subj);
                        }
                    } else if (is_member_class) { /* membership probe */
                        if (fresh) bytes += 4 * clog2(deg + 1);
                        for (size_t i = lo; i < hi; i++)
                            if (o->tup[i].stype == a->stype && o->tup[i].srel == a->srel && o->tup[i].subj == s.sid && tup_live(o, &o->tup[i])) found = 1;
                    }
                }
            } else {
                const expr_t *stack[256];
                int sp = 0;
                stack[sp++] = r->expr;
                while (sp) {
                    const expr_t *e = stack[--sp];
                    if (e->kind == EX_UNION || e->kind == EX_INTERSECT || e->kind == EX_EXCLUDE) { stack[sp++] = e->l; stack[sp++] = e->r; }
                    else if (e->kind == EX_REF) PUSH_NEXT(st.type, rel_index(&o->types[st.type], e->a), st.id);
                    else if (e->kind == EX_ARROW || e->kind == EX_ARROW_ALL) {
                        int ts = rel_index(&o->types[st.type], e->a);
                        rel_t *tr = &o->types[st.type].rels[ts];
                        size_t lo, hi;
                        row_range(o, st.type, ts, st.id, &lo, &hi);
                        for (int k = 0; k < tr->nallowed; k++) {
                            const allowed_t *a = &tr->allowed[k];
                            uint64_t rowkey = ((uint64_t)st.type << 56) ^ ((uint64_t)ts << 48) ^ ((uint64_t)k << 40) ^ ((uint64_t)st.id);
                            rowkey ^= 0x8000000000000000ull;
                            int fresh = set_add(&seen_rows, rowkey);
                            if (fresh) bytes += 8;
                            int target = rel_index(&o->types[a->stype], e->b);
                            for (size_t i = lo; i < hi; i++) {
                                const tuple_t *tp = &o->tup[i];
                                if (tp->stype != a->stype || tp->srel != a->srel || !tup_live(o, tp)) continue;
                                if (fresh) bytes += 4;
                                if (target >= 0) PUSH_NEXT(tp->stype, target, tp->subj);
                            }
                        }
                    }
                }
            }
        }
        state_t *tmp = front; front = next; next = tmp;
        fn = nn;
        if (lvl_bytes) lvl_bytes[level] += bytes - charged;
        charged = bytes;
    }
#undef PUSH_NEXT
    free(front); free(next); free(seen_states.k); free(seen_rows.k);
    if (result) *result = found ? ORC_PERM_HAS : ORC_PERM_NO;
    return bytes;
}

uint64_t orc_check_bytes(orc_t *o, int rtype, int perm, uint32_t res, int stype, int srel, uint32_t subj, int *result) {
    ensure_sorted(o);
    return check_bytes_core(o, rtype, perm, res, stype, srel, subj, result, NULL, NULL);
}

/* The byte model over a WHOLE batch, split over host threads (the model only reads the sorted tuple table).
 * Returns the batch's algorithmic bytes; lvl_bytes / lvl_states: [ORC_MAX_DEPTH + 1] totals per level (may be NULL). */
typedef struct {
    orc_t *o;
    size_t lo, hi;
    int rtype, perm, stype, srel;
    const uint32_t *res, *subj;
    uint64_t total, lvl_bytes[ORC_MAX_DEPTH + 1], lvl_states[ORC_MAX_DEPTH + 1];
} bytes_job_t;
static void *bytes_run(void *arg) {
    bytes_job_t *j = arg;
    for (size_t i = j->lo; i < j->hi; i++)
        j->total += check_bytes_core(j->o, j->rtype, j->perm, j->res[i], j->stype, j->srel, j->subj[i], NULL, j->lvl_bytes, j->lvl_states);
    return NULL;
}
uint64_t orc_check_bytes_bulk_mt(orc_t *o, int nthreads, size_t n, int rtype, int perm, const uint32_t *res, int stype, int srel, const uint32_t *subj,
                                 uint64_t *lvl_bytes, uint64_t *lvl_states) {
    ensure_sorted(o);
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n && n) nthreads = (int)n;
    bytes_job_t *jobs = calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *th = calloc((size_t)nthreads, sizeof *th);
    for (int t = 0; t < nthreads; t++) {
        bytes_job_t *j = &jobs[t];
        j->o = o;
        j->lo = n * (size_t)t / (size_t)nthreads;
        j->hi = n * (size_t)(t + 1) / (size_t)nthreads;
        j->rtype = rtype; j->perm = perm; j->stype = stype; j->srel = srel;
        j->res = res; j->subj = subj;
        pthread_create(&th[t], NULL, bytes_run, j);
    }
    uint64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        total += jobs[t].total;
        for (int l = 0; l <= ORC_MAX_DEPTH; l++) {
            if (lvl_bytes) lvl_bytes[l] += jobs[t].lvl_bytes[l];
            if (lvl_states) lvl_states[l] += jobs[t].lvl_states[l];
        }
    }
    free(jobs); free(th);
    return total;
}

/* multi-threaded helper for the cpu_baseline leg is deliberately absent: the
 * oracle is a scalar single-thread port ("cores": 1). */


/* ------------------------------------------------------------------ tuned CPU Check (bench.py `cpu_baseline.tuned`)
 * the recursive evaluator above is the CHECKER -- written to be obviously right, it binary-searches a 24-byte tuple array
 * three times per row and memoises in a hash table -- and a GPU / CPU ratio against it says nothing.  This is the same decision procedure written to be
 * fast on a CPU, for the schemas the BASELINE configurations use (unions, arrows, computed usersets, userset subjects, `nil`; anything with `&`, `-`,
 * `.all()` or a wildcard relationship is REFUSED: orc_tuned_supported() == 0):
 *   - a row is found through an index: rowptr[type][relation][resource id] = its first tuple (one load, no search);
 *   - a Check is evaluated LEVEL-SYNCHRONOUSLY, as the device does: the frontier of (type, relation, id) states at dispatch level L is expanded into
 *     level L + 1, identical states of a level are merged (their subtrees are identical), the walk stops at the first membership hit;
 *   - the batch is handed out in chunks of 256 requests to `nthreads` threads (requests differ 100-fold in work: a static split leaves threads idle).
 * Semantics = check_rel()'s: HAS if some chain of <= 50 dispatches reaches a relationship naming the subject (or the subject itself, as a userset);
 * else ERR if some chain needs a 51st dispatch; else NO.  Every caller asserts the answers EQUAL the recursive evaluator's (tests/test_oracle_cross.py,
 * bench.py).  Test infrastructure like the rest of this file: never linked into the product. */
static int tn_expr_ok(const expr_t *e) {
    if (!e) return 1;
    if (e->kind == EX_INTERSECT || e->kind == EX_EXCLUDE || e->kind == EX_ARROW_ALL) return 0;
    return tn_expr_ok(e->l) && tn_expr_ok(e->r);
}
int orc_tuned_supported(orc_t *o) {
    ensure_sorted(o);
    for (int t = 0; t < o->ntypes; t++)
        for (int r = 0; r < o->types[t].nrels; r++)
            if (o->types[t].rels[r].is_perm && !tn_expr_ok(o->types[t].rels[r].expr)) return 0;
    for (size_t i = 0; i < o->ntup; i++)
        if (o->tup[i].srel == WILDCARD) return 0;
    return 1;
}
int orc_tuned_build(orc_t *o) {
    if (!orc_tuned_supported(o)) return -1;
    if (o->tn_ready) return 0;
    int maxrels = 1;
    for (int t = 0; t < o->ntypes; t++) if (o->types[t].nrels > maxrels) maxrels = o->types[t].nrels;
    o->tn_maxrels = maxrels;
    o->tn_rowptr = calloc((size_t)o->ntypes * (size_t)maxrels, sizeof *o->tn_rowptr);
    o->tn_maxid = calloc((size_t)o->ntypes * (size_t)maxrels, sizeof *o->tn_maxid);
    for (size_t i = 0; i < o->ntup;) { /* the tuples are sorted by (type, relation, resource, ...): one run per relation */
        const tuple_t *a = &o->tup[i];
        size_t j = i;
        while (j < o->ntup && o->tup[j].rtype == a->rtype && o->tup[j].rel == a->rel) j++;
        const uint32_t maxid = o->tup[j - 1].res;
        uint32_t *rp = malloc(((size_t)maxid + 2) * sizeof *rp);
        size_t k = i;
        for (uint32_t id = 0; id <= maxid; id++) {
            rp[id] = (uint32_t)k;
            while (k < j && o->tup[k].res == id) k++;
        }
        rp[maxid + 1] = (uint32_t)j;
        o->tn_rowptr[(size_t)a->rtype * (size_t)maxrels + a->rel] = rp;
        o->tn_maxid[(size_t)a->rtype * (size_t)maxrels + a->rel] = maxid;
        i = j;
    }
    o->tn_ready = 1;
    return 0;
}
static inline void tn_row(const orc_t *o, int type, int rel, uint32_t id, size_t *lo, size_t *hi) {
    const size_t k = (size_t)type * (size_t)o->tn_maxrels + (size_t)rel;
    const uint32_t *rp = o->tn_rowptr[k];
    if (!rp || id > o->tn_maxid[k]) { *lo = *hi = 0; return; }
    *lo = rp[id];
    *hi = rp[id + 1];
}
typedef struct { uint64_t *a, *b; size_t na, nb, cap; } tn_scratch_t;
static inline void tn_push(tn_scratch_t *s, int type, int rel, uint32_t id) {
    if (s->nb == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 1024;
        s->a = realloc(s->a, s->cap * sizeof(uint64_t));
        s->b = realloc(s->b, s->cap * sizeof(uint64_t));
    }
    s->b[s->nb++] = ((uint64_t)(unsigned)type << 48) | ((uint64_t)(unsigned)rel << 32) | id;
}
static int u64_cmp(const void *x, const void *y) { uint64_t a = *(const uint64_t *)x, b = *(const uint64_t *)y; return a < b ? -1 : a > b; }
/* the children of permission expression e on (type, id): states of the NEXT dispatch level */
static void tn_expr(const orc_t *o, tn_scratch_t *s, int type, const expr_t *e, uint32_t id) {
    const type_t *t = &o->types[type];
    switch (e->kind) {
    case EX_UNION: tn_expr(o, s, type, e->l, id); tn_expr(o, s, type, e->r, id); break;
    case EX_REF: tn_push(s, type, rel_index(t, e->a), id); break;
    case EX_ARROW: {
        size_t lo, hi;
        tn_row(o, type, rel_index(t, e->a), id, &lo, &hi);
        for (size_t i = lo; i < hi; i++) {
            const tuple_t *tp = &o->tup[i];
            if (!tup_live(o, tp)) continue;
            const int tr = rel_index(&o->types[tp->stype], e->b);
            if (tr >= 0) tn_push(s, tp->stype, tr, tp->subj);
        }
        break;
    }
    default: break; /* nil */
    }
}
static int tn_check(const orc_t *o, tn_scratch_t *s, int rtype, int perm, uint32_t res, const subject_t *sub) {
    s->na = 0;
    s->nb = 0;
    tn_push(s, rtype, perm, res);
    for (int level = 1; level <= ORC_MAX_DEPTH; level++) {
        /* level L = what tn_push collected: merge identical states, then expand every one */
        { uint64_t *x = s->a; s->a = s->b; s->b = x; s->na = s->nb; s->nb = 0; }
        if (s->na > 1) {
            qsort(s->a, s->na, sizeof(uint64_t), u64_cmp);
            size_t w = 1;
            for (size_t i = 1; i < s->na; i++) if (s->a[i] != s->a[w - 1]) s->a[w++] = s->a[i];
            s->na = w;
        }
        if (!s->na) return R_NO;
        for (size_t q = 0; q < s->na; q++) {
            const int type = (int)(s->a[q] >> 48), rel = (int)((s->a[q] >> 32) & 0xFFFF);
            const uint32_t id = (uint32_t)s->a[q];
            if (sub->stype == type && sub->srel == (unsigned)rel && sub->sid == id) return R_HAS; /* the subject itself, as a userset */
            const rel_t *r = &o->types[type].rels[rel];
            if (r->is_perm) { tn_expr(o, s, type, r->expr, id); continue; }
            size_t lo, hi;
            tn_row(o, type, rel, id, &lo, &hi);
            if (lo < hi) {
                const size_t p = row_find(o, lo, hi, sub->stype, sub->srel, sub->sid);
                if (p < hi && tup_live(o, &o->tup[p])) return R_HAS;
                for (size_t i = lo; i < hi; i++) { /* userset subjects: one dispatch level further */
                    const tuple_t *tp = &o->tup[i];
                    if (tp->srel == ELLIPSIS || !tup_live(o, tp)) continue;
                    tn_push(s, tp->stype, (int)tp->srel, tp->subj);
                }
            }
        }
    }
    return s->nb ? R_ERR : R_NO; /* a 51st dispatch would be needed */
}
typedef struct {
    const orc_t *o;
    size_t n, *next;
    pthread_mutex_t *mu;
    int rtype, perm, stype, srel;
    const uint32_t *res, *subj;
    uint8_t *out;
    int32_t *err;
} tn_job_t;
static void *tn_run(void *p) {
    tn_job_t *j = (tn_job_t *)p;
    tn_scratch_t s = {0};
    for (;;) {
        pthread_mutex_lock(j->mu);
        const size_t lo = *j->next;
        *j->next = lo + 256;
        pthread_mutex_unlock(j->mu);
        if (lo >= j->n) break;
        const size_t hi = lo + 256 < j->n ? lo + 256 : j->n;
        for (size_t i = lo; i < hi; i++) {
            const subject_t sub = {j->stype, j->srel < 0 ? ELLIPSIS : (unsigned)j->srel, j->subj[i]};
            const int r = tn_check(j->o, &s, j->rtype, j->perm, j->res[i], &sub);
            j->out[i] = r == R_HAS ? ORC_PERM_HAS : (r == R_ERR ? ORC_PERM_UNSPEC : ORC_PERM_NO);
            if (j->err) j->err[i] = r == R_ERR ? ORC_ERR_DEPTH : 0;
        }
    }
    free(s.a);
    free(s.b);
    return NULL;
}
/* -1: the schema / data are outside what the tuned evaluator takes (the caller keeps the recursive one) */
int orc_tuned_check_bulk_ids_mt(orc_t *o, int nthreads, size_t n, int rtype, int perm, const uint32_t *res, int stype, int srel, const uint32_t *subj,
                                uint8_t *out, int32_t *err) {
    if (orc_tuned_build(o)) return -1;
    if (nthreads < 1) nthreads = 1;
    size_t next = 0;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    tn_job_t job = {o, n, &next, &mu, rtype, perm, stype, srel, res, subj, out, err};
    pthread_t *th = calloc((size_t)nthreads, sizeof *th);
    for (int t = 1; t < nthreads; t++) pthread_create(&th[t], NULL, tn_run, &job);
    tn_run(&job);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    return 0;
}
