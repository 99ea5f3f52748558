/*
 * aclgpu.h -- C ABI of libaclgpu.so, the MI355X-native batched ACL-check engine.
 *
 * This is the drop-in boundary for ONE path of authzed/spicedb-kubeapi-proxy: the
 * embedded-SpiceDB Check / Filter (LookupResources) path.  The reference has no
 * C ABI (it is pure Go, CGO_ENABLED=0); the seam it replaces is the Go interface
 * v1.PermissionsServiceClient held in proxy.Options.PermissionsClient
 * (reference pkg/proxy/options.go:82, auto-constructed only when nil at
 * options.go:371-377).  Every entry point below names the reference call site
 * whose request it answers; INTEGRATION.md shows the cgo binding that implements
 * the Go interface on top of these symbols.
 *
 * Conventions
 *  - plain C types only; the caller owns every input for the duration of the call
 *    and provides every output buffer; engine-owned strings stay valid until the
 *    next mutating call on the same handle or acl_close().
 *  - every function returns 0 (ACL_OK) or a gRPC status code (the codes the
 *    reference inspects: codes.InvalidArgument pkg/authz/distributedtx/workflow.go:115,
 *    precondition / already-exists failures activity.go:62-74); the message is
 *    available from acl_last_error() (thread-local).
 *  - all entry points are thread-safe and concurrent (the proxy calls the seam from arbitrary
 *    goroutines: pkg/authz/check.go:77-93, responsefilterer.go:165): evaluations share the
 *    relationship store and the HBM snapshot and run on separate device contexts (HIP
 *    streams); writes are exclusive and visible to every call that starts after them.
 *  - there is NO CPU evaluation path in this library: if no gfx950 device is
 *    usable acl_open() fails with ACL_ERR_UNAVAILABLE.
 */
#ifndef ACLGPU_H
#define ACLGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct acl_engine acl_engine_t;

/* gRPC status codes used as return values */
enum {
    ACL_OK = 0,
    ACL_ERR_CANCELLED = 1,           /* codes.Canceled: the caller's cancel flag was raised (responsefilterer.go:168-170) */
    ACL_ERR_INVALID_ARGUMENT = 3,    /* codes.InvalidArgument   */
    ACL_ERR_DEADLINE_EXCEEDED = 4,   /* codes.DeadlineExceeded: acl_call_opts_t.timeout_ns elapsed (responsefilterer.go:44,196-204) */
    ACL_ERR_NOT_FOUND = 5,           /* codes.NotFound          */
    ACL_ERR_ALREADY_EXISTS = 6,      /* codes.AlreadyExists: CREATE of an existing relationship */
    ACL_ERR_PERMISSION_DENIED = 7,   /* codes.PermissionDenied: acl_prefilter_response on a single object outside the allowed set ("unauthorized") */
    ACL_ERR_RESOURCE_EXHAUSTED = 8,  /* codes.ResourceExhausted: frontier capacity exceeded */
    ACL_ERR_FAILED_PRECONDITION = 9, /* codes.FailedPrecondition: precondition failed, unknown type/relation */
    ACL_ERR_OUT_OF_RANGE = 11,       /* codes.OutOfRange: watch cursor older than the retained change feed */
    ACL_ERR_INTERNAL = 13,           /* codes.Internal (HIP runtime failure) */
    ACL_ERR_UNAVAILABLE = 14,        /* codes.Unavailable: no usable GPU */
    ACL_ERR_DEPTH = 100              /* per-item: max dispatch depth (50) exceeded, pkg/spicedb/spicedb.go:34 */
};

/* authzed.api.v1.CheckPermissionResponse.Permissionship (authzed-go v1.10.0, go.mod:6);
 * only HAS_PERMISSION allows: pkg/authz/check.go:63, postfilter.go:169, watch.go:104 */
enum { ACL_PERM_UNSPECIFIED = 0, ACL_PERM_NO_PERMISSION = 1, ACL_PERM_HAS_PERMISSION = 2, ACL_PERM_CONDITIONAL = 3 };
/* authzed.api.v1.RelationshipUpdate.Operation (used by pkg/authz/distributedtx/activity.go:47-77) */
enum { ACL_OP_CREATE = 1, ACL_OP_TOUCH = 2, ACL_OP_DELETE = 3 };
/* authzed.api.v1.Precondition.Operation (workflow.go:452-462) */
enum { ACL_PRE_MUST_NOT_MATCH = 1, ACL_PRE_MUST_MATCH = 2 };

#define ACL_NO_RELATION 0xFFFFu /* "no subject relation" (ellipsis) in interned items */

typedef struct {
    int32_t device;              /* HIP device ordinal; -1 = current (LOCAL_RANK for one-process-per-GPU) */
    uint64_t frontier_entries;   /* capacity of EACH of the two frontier buffers, in 16-byte entries; 0 = default (32 M: 512 MiB each, per context) */
    uint32_t max_sub_batch;      /* Check items evaluated per device pass; 0 = default */
    uint32_t flags;              /* ACL_FLAG_* */
    uint32_t contexts;           /* evaluations that may be in flight on the device at once (each has its own HIP stream,
                                    frontier and staging buffers; created on demand); 0 = default (4), max 16 */
    uint32_t reserved;
} acl_config_t;
/* open only the relationship store (writes, reads, preconditions) without touching a GPU: every entry
 * point that evaluates permissions then fails with ACL_ERR_UNAVAILABLE.  For tooling and CPU-side tests. */
#define ACL_FLAG_STORE_ONLY 1u
/* CheckBulkPermissions: an item whose fields fail the API's patterns carries InvalidArgument in ITS pair instead of failing the whole call.
 * The validation rules are a restatement from memory of the authzed API's `validate` tags (csrc/validate.hpp: unverified until the Go reference
 * harness replays them), and the default -- as embedded SpiceDB does -- fails the WHOLE request, which the reference turns into a blanket denial
 * (pkg/authz/check.go:48-52) or a failed list (postfilter.go:134-137).  An operator who meets that for ids the real engine would take can
 * confine the failure to the offending pairs with this flag. */
#define ACL_FLAG_PER_ITEM_VALIDATION 2u
/* LookupResources over a permission that holds an intersection / exclusion confirms its candidates with a forward Check; a candidate whose Check
 * ERRS (a branch beyond the dispatch depth) fails the call with that item's code -- the reference's stream ends at the first Recv error and the list
 * request with it (pkg/authz/lookups.go:75-83, responsefilterer.go:196-204).  This flag keeps the lenient form of rounds 4-5: such candidates are
 * dropped from the answer and the call succeeds. */
#define ACL_FLAG_LENIENT_LOOKUP 4u
/* create all `contexts` evaluation contexts at acl_open instead of on demand: a server pays their allocation (frontier buffers, pinned staging:
 * milliseconds each) before its first request, not inside the first requests that find the pool busy. */
#define ACL_FLAG_EAGER_CONTEXTS 8u

/* replaces spicedb.NewServer (pkg/spicedb/spicedb.go:18-71): builds the engine. */
int acl_open(const acl_config_t *cfg, acl_engine_t **out);
/* ONE engine in front of several GPUs of this process: one relationship store, one set of name tables, one HBM snapshot PER DEVICE of
 * `devices` (HIP ordinals; a device may be listed more than once: logical replicas).  Every snapshot update reaches every replica before the
 * write / read that caused it returns, evaluations are spread over the replicas, so the handle behaves as the reference's single
 * PermissionsClient does (pkg/proxy/options.go:371-377; the dual-write worker shares it, pkg/proxy/server.go:136-153): read-your-writes for
 * the whole process.  Entry points that take DEVICE pointers run on the replica of the pointers' device.  acl_open honours
 * ACL_DEVICES="0,1,2,3" the same way.  The acl_shard_* entry points use the first replica only. */
int acl_open_replicas(const acl_config_t *cfg, const int32_t *devices, uint32_t n_devices, acl_engine_t **out);
/* returns the number of replicas; fills (up to cap) how many evaluations each has been handed since open and its HIP ordinal.  The handle
 * still is the reference's ONE client (options.go:371-377): this only shows how the calls were spread. */
int acl_replica_calls(acl_engine_t *h, uint64_t *calls_out, int32_t *devices_out, uint32_t cap);
/* no other call on the handle may be in flight (a poller blocked in acl_check_completions included: stop it first) */
void acl_close(acl_engine_t *h);
const char *acl_last_error(void);

/* bootstrap: schema text + relationship lines, as in pkg/spicedb/bootstrap.yaml:1-40
 * (spicedb.go:19-24).  rels_utf8 may be NULL; one `type:id#rel@type:id[#rel]` per line
 * (grammar pkg/rules/rules.go:1053-1055). Replaces any previous schema and data. */
int acl_load_bootstrap(acl_engine_t *h, const char *schema_utf8, size_t schema_len, const char *rels_utf8, size_t rels_len);
/* ... and the bootstrap FILE form: YAML documents `{schema: <text>, relationships: <lines>}` as the reference's embedded default
 * (pkg/spicedb/bootstrap.yaml), a file named by the endpoint URL (pkg/proxy/options.go:313-316, spicedb.go:22-23) or a byte map
 * (spicedb.go:19-21) hold them.  Several documents (`---`) are merged.  The YAML subset is the one such files use (top-level keys, block or
 * one-line scalars; csrc/bootstrap_yaml.cpp); anything else is INVALID_ARGUMENT, never a guess. */
int acl_load_bootstrap_yaml(acl_engine_t *h, const char *yaml_utf8, size_t len);

/* ---- identifiers (pre-interned fast path) ---- */
int acl_type_id(acl_engine_t *h, const char *type);                 /* -1 if unknown */
int acl_relation_id(acl_engine_t *h, int type, const char *name);   /* relation or permission; -1 if unknown */
int acl_intern(acl_engine_t *h, int type, const char *object_id, uint32_t *id_out); /* creates if missing */
int acl_find(acl_engine_t *h, int type, const char *object_id, uint32_t *id_out);   /* ACL_ERR_NOT_FOUND if missing */
const char *acl_object_name(acl_engine_t *h, int type, uint32_t id); /* NULL for anonymous/unknown ids.  The pointer is only good while the id keeps its
                                                                      * name: ids of objects that take part in no relationship are recycled (a quarantine after
                                                                      * they were last handed out), and a recycled id's name bytes are overwritten.  Prefer: */
/* Copies the name of `id` into buf (NUL-terminated, at most cap bytes incl. the NUL) under the names lock and returns its length (which may exceed cap - 1:
 * the copy is then truncated); -1 for anonymous / unknown ids.  What the cgo shim's LookupResources stream calls per result id (lookups.go:75-83). */
int64_t acl_object_name_copy(acl_engine_t *h, int type, uint32_t id, char *buf, size_t cap);
/* The names of the objects whose bits are set in a LookupResources bitmap (lookups.go:75-83: the reference drains one message per result), a BLOCK per call: from bit
 * *cursor on (0 at first), up to max_names names back to back into buf (cap >= 1024 bytes, not NUL-terminated; ends[k] = offset one past name k; an id without a
 * name gets an empty one).  *cursor is left at the first bit not reported yet, *n_out = names written; *n_out == 0: the bitmap is exhausted. */
int acl_bitmap_names(acl_engine_t *h, int type, const uint32_t *bitmap, size_t words, uint64_t *cursor, char *buf, size_t cap, uint32_t *ends, size_t max_names, size_t *n_out);
uint32_t acl_object_count(acl_engine_t *h, int type);                /* size of the type's dense id space */

/* ---- relationship store: the write side of the seam ---- */
typedef struct {
    const char *resource_type, *resource_id, *relation;
    const char *subject_type, *subject_id, *subject_relation; /* NULL or "" = none */
    int64_t expires_at;                                      /* unix seconds; 0 = never (spicedb.go:60) */
} acl_relationship_t;
typedef struct { int32_t op; acl_relationship_t rel; } acl_update_t;
typedef struct {
    int32_t op;                /* ACL_PRE_*; ignored by read/delete */
    const char *resource_type; /* required */
    const char *resource_id;   /* NULL/"" = any */
    const char *relation;      /* NULL/"" = any */
    const char *subject_type;  /* NULL = no subject filter */
    const char *subject_id;    /* NULL/"" = any */
    const char *subject_relation; /* NULL = any; "" = only "no relation"; else exact */
} acl_filter_t;

/* WriteRelationships (activity.go:60): atomic; preconditions see the pre-write state;
 * <=1000 updates and <=1000 preconditions (spicedb.go:35-36).  *revision_out = ZedToken. */
int acl_write(acl_engine_t *h, const acl_update_t *updates, int n_updates, const acl_filter_t *preconditions, int n_pre,
              uint64_t *revision_out);
/* DeleteRelationships (e2e/util_test.go:66) */
int acl_delete_by_filter(acl_engine_t *h, const acl_filter_t *filter, uint64_t *n_deleted, uint64_t *revision_out);
/* ... with DeleteRelationshipsRequest.OptionalPreconditions: evaluated against the pre-delete state, atomically with it */
int acl_delete_by_filter_pre(acl_engine_t *h, const acl_filter_t *filter, const acl_filter_t *preconditions, int n_pre, uint64_t *n_deleted,
                             uint64_t *revision_out);
/* ReadRelationships (activity.go:107, e2e/util_test.go:27): cb per matching relationship */
typedef void (*acl_read_cb)(void *user, const acl_relationship_t *rel);
int acl_read(acl_engine_t *h, const acl_filter_t *filter, acl_read_cb cb, void *user);
/* bulk load with caller-chosen dense numeric ids (ImportBulkRelationships analogue; TOUCH semantics) */
int acl_add_edges(acl_engine_t *h, int rtype, int relation, int stype, int srel /* -1 none; -2: `stype:*` wildcard relationships (subject_ids ignored) */, size_t n, const uint32_t *resource_ids,
                  const uint32_t *subject_ids);
uint64_t acl_revision(acl_engine_t *h);
/* test clock for relationship expiration; 0 = wall clock */
int acl_set_now(acl_engine_t *h, int64_t unix_seconds);
/* build + upload the HBM snapshot now (otherwise done lazily by the next read) */
int acl_snapshot(acl_engine_t *h);

/* ---- Check: CheckBulkPermissions (check.go:48, postfilter.go:134), CheckPermission (watch.go:50) ---- */
typedef struct {
    const char *resource_type, *resource_id, *permission;
    const char *subject_type, *subject_id, *subject_relation; /* NULL or "" = none (check.go:36) */
} acl_check_item_t;
/* order preserving: perm_out[i] / err_out[i] answer items[i] (check.go:54-57).
 * err_out[i] != 0 is that pair's Error (check.go:55); perm_out[i] is then UNSPECIFIED. */
int acl_check_bulk(acl_engine_t *h, const acl_check_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out);
/* The same request with {pointer, length} fields: no NUL behind the bytes, no strlen per field.  A cgo shim points these at the Go strings'
 * own bytes (pinned for the call) instead of C.CString-copying six strings per item; an absent subject relation is {NULL, 0}. */
typedef struct {
    const char *p;
    size_t n;
} acl_str_t;
typedef struct {
    acl_str_t resource_type, resource_id, permission, subject_type, subject_id, subject_relation;
} acl_check_item_v_t;
int acl_check_bulk_v(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, uint8_t *perm_out, int32_t *err_out);

/* 16-byte interned request (SURVEY 8(a) a3 "interned it is 16 B") */
typedef struct {
    uint16_t resource_type, permission; /* acl_type_id / acl_relation_id */
    uint32_t resource_id;
    uint16_t subject_type, subject_relation; /* ACL_NO_RELATION = none */
    uint32_t subject_id;
} acl_item_t;
int acl_check_bulk_ids(acl_engine_t *h, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out);
/* Names -> ids in bulk, no device pass (also on a store-only engine): out[i] is items[i] as acl_check_bulk_ids takes it -- exactly what
 * acl_check_bulk_v resolves before it asks the device, so acl_check_bulk_ids(out) answers as acl_check_bulk_v(items) would.  A shim that is asked about the
 * same objects again (the user of a request, check.go:17-72; the namespaces of a list, postfilter.go:88-119) resolves once and keeps the ids.
 * Object names the graph does not know resolve to ids without relationships; err_out[i] != 0 (unknown type / permission: FAILED_PRECONDITION, an empty
 * or ill-formed field: INVALID_ARGUMENT -- always per item here) makes out[i] an item that every Check answers UNSPECIFIED with an error.
 * An id stays its object's while the object has relationships, and for the length of a recycling quarantine (30 s) after this call otherwise.
 * KEEPING the ids: an object name no table knows resolves to a sentinel id >= ACL_UNKNOWN_ID_MIN, which stays "nobody" after the object is
 * created -- items that carry one must NOT be cached; and a cached id of an object that may lose all its relationships must be resolved again (or touched
 * through acl_find) within the quarantine, or it may come to name another object.  Ids of objects that keep relationships are stable. */
#define ACL_UNKNOWN_ID_MIN 0xFFFFFFF0u
int acl_resolve_bulk_v(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, acl_item_t *out, int32_t *err_out);

/* Cancellation / deadline of one call -- the C side of a Go context.Context.  The reference runs LookupResources on the
 * HTTP request's ctx and abandons it when that is cancelled (responsefilterer.go:165-170); the prefilter join gives up
 * after 10 s (responsefilterer.go:44,196-204).  The shim points `cancel` at an int32 it sets from a goroutine watching
 * ctx.Done(); the engine polls it while the call waits for a device context / the micro-batcher and between level
 * bursts of the walk, and returns ACL_ERR_CANCELLED / ACL_ERR_DEADLINE_EXCEEDED.  NULL opts = neither. */
typedef struct {
    const volatile int32_t *cancel; /* NULL or a flag: != 0 abandons the call */
    int64_t timeout_ns;             /* <= 0: none; else relative to the call's start */
} acl_call_opts_t;
int acl_check_bulk_ids_opts(acl_engine_t *h, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *opts);
/* acl_check_bulk_v under the caller's context (reference pkg/authz/check.go:48, postfilter.go:134: CheckBulkPermissions(ctx, ...)): a raised
 * cancel flag or a passed deadline ends the call with CANCELLED / DEADLINE_EXCEEDED while it waits for an evaluation context. */
int acl_check_bulk_v_opts(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *opts);
/* The same request PACKED (round 6): the call's DISTINCT strings once, back to back, and six dictionary indices per item.  A shim that walks a kube list
 * copies every string once anyway; written into one buffer an item is 24 bytes instead of six {pointer, length} views, the fields every pair of a
 * PostFilter call shares (type, permission, the user: pkg/authz/postfilter.go:88-119) are ONE dictionary entry -- found equal by index --, and a name that
 * many items carry is resolved once per call.  String s is bytes[offsets[s], offsets[s + 1]); item i is items[6 i .. 6 i + 5] = resource type, resource id,
 * permission, subject type, subject id, subject relation; ACL_PACKED_NONE = the member is absent (an empty request's nil Resource: that pair is answered
 * InvalidArgument as for {NULL, 0} views, pkg/proxy/options_test.go:101-102).  Any other index >= n_strings fails the call with INVALID_ARGUMENT. */
#define ACL_PACKED_NONE 0xFFFFFFFFu
typedef struct {
    const char *bytes;
    const uint32_t *offsets; /* [n_strings + 1], ascending */
    uint32_t n_strings;
    uint32_t reserved;
    const uint32_t *items;   /* [n_items][6] */
    size_t n_items;
} acl_packed_request_t;
/* opts: the caller's context (cancel flag / deadline), as acl_check_bulk_v_opts takes it; NULL = neither. */
int acl_check_bulk_packed(acl_engine_t *h, const acl_packed_request_t *req, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *opts);

/* Pipelined form of acl_check_bulk_ids for hosts that cannot park a thread per call: submit returns at once, the batch is answered as a
 * whole blocking call on one of the engine's pool workers (one worker per evaluation context); acl_ticket_wait blocks until perm_out /
 * err_out are filled, returns the call's status and frees the ticket.  Buffers must stay valid until the wait returns.  A ticket nobody
 * has waited for yet pins nothing: the holder of tickets may make any other call on the engine (writes included) before it waits.
 * Every batch -- a blocking caller's or a ticket's -- is answered by the kernel itself across PCIe: it reads the items from, and writes
 * the answers to, the host buffers (pinned: in place; else through the context's pinned staging), in one launch, or -- beyond what one
 * launch takes -- in sub-passes on two streams.  No copies, no turn-taking: concurrent calls overlap on the chip (three callers 1.22-1.29 G
 * decisions/s on C4, one caller 0.85 G; profiles/r04_host_split.txt).  Go callers simply block goroutines in acl_check_bulk_ids. */
typedef struct acl_ticket acl_ticket_t;
int acl_check_bulk_ids_submit(acl_engine_t *h, const acl_item_t *items, size_t n, uint8_t *perm_out, int32_t *err_out, acl_ticket_t **ticket_out);
int acl_ticket_wait(acl_engine_t *h, acl_ticket_t *ticket);
/* Page-locked host memory for request / answer arrays: buffers from here are read / written by the kernels, or DMA'd, directly by the
 * host-buffer entry points; any other host pointer is staged through the context's own pinned buffers (one extra memcpy). */
int acl_host_alloc(acl_engine_t *h, size_t bytes, void **out);
int acl_host_free(acl_engine_t *h, void *p);
/* same, but items / outputs are DEVICE pointers (HBM-resident batch).  The inputs must be complete before the call
 * (they are read on one of the engine's own streams); the answers are complete when it returns.  err_out may be NULL */
int acl_check_bulk_ids_device(acl_engine_t *h, const void *d_items, size_t n, void *d_perm_out, void *d_err_out);
void *acl_stream(acl_engine_t *h); /* hipStream_t of the engine's first evaluation context */
int acl_sync(acl_engine_t *h);     /* waits for every evaluation context's stream */

/* ---- Filter: LookupResources (lookups.go:49-65) ----
 * Result = dense bitmap over the resource type's local ids (bit id set <=> HAS_PERMISSION);
 * the Go side intersects it with the kube list (lookups.go:25-36, responsefilterer.go:349-415). */
int acl_lookup_resources(acl_engine_t *h, const char *resource_type, const char *permission, const char *subject_type,
                         const char *subject_id, const char *subject_relation, uint32_t *bitmap_out, size_t bitmap_words,
                         uint64_t *count_out);
int acl_lookup_resources_ids(acl_engine_t *h, int rtype, int permission, int stype, int srel /* -1 none */, uint32_t subject_id,
                             uint32_t *bitmap_out, size_t bitmap_words, uint64_t *count_out);
/* Engine-owned result: *bitmap_out (release with acl_free) is sized by the engine when the walk runs, so objects interned
 * by a racing WriteRelationships can never make a caller-sized buffer "too small".  Rides the micro-batcher like
 * acl_lookup_one when it is running.  opts may be NULL. */
int acl_lookup_resources_alloc(acl_engine_t *h, const char *resource_type, const char *permission, const char *subject_type,
                               const char *subject_id, const char *subject_relation, const acl_call_opts_t *opts, uint32_t **bitmap_out,
                               size_t *bitmap_words_out, uint64_t *count_out);
void acl_free(void *p);
/* batched form: n subjects of one (stype, srel) against one (rtype, permission); bitmaps_out is n * bitmap_words */
int acl_lookup_resources_batch(acl_engine_t *h, int rtype, int permission, int stype, int srel, const uint32_t *subject_ids, size_t n,
                               uint32_t *bitmaps_out, size_t bitmap_words, uint64_t *counts_out);

/* ---- the callers either side of the kernels ----
 * PostFilter: filterItemsWithBulkPermissions (postfilter.go:58-182).  The K list items' resolved pairs are ONE bulk
 * check; pairs [item_off[i], item_off[i+1]) belong to list item i (itemToRequestMap, postfilter.go:65,117-119);
 * keep_out[i] = 1 iff all of them are HAS_PERMISSION without error (postfilter.go:152-178); an item without pairs
 * is kept (postfilter.go:145-150).  K bytes come back; the *_device form does the AND on the device (its answers never leave the HBM). */
int acl_check_bulk_keep(acl_engine_t *h, const acl_check_item_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out);
int acl_check_bulk_keep_ids(acl_engine_t *h, const acl_item_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out);
/* ... with {pointer, length} fields / a packed request.  Round 6: the reference's PostFilter names ONE subject for all K x F pairs (postfilter.go:67-119:
 * every template is resolved for the requesting user).  When every pair of the call shares (resource type, permission, subject) -- a plain subject, a
 * permission without `&` / `-` -- the call is answered by ONE reverse walk from that subject (the LookupResources kernel) and K bit tests: the pairs' resource
 * names are hashed and tested against the few ALLOWED names first, so a name the user may not see never touches the type's name table (the string path's
 * cost is that table: one DRAM miss per name).  K items x F templates (every item F pairs, pair j from template j, F = 2 ... 4): one walk per template under one
 * evaluation, an item kept when every template keeps it.  Any other call -- subjects or permissions that differ, a userset subject, an item the API would refuse --
 * takes the forward path; the keep mask and the call's error are the same either way (tests/test_callers_gpu.py compares the two routes and the oracle).
 * acl_check_bulk / _v / _packed (CheckBulkPermissions itself, what the unpatched proxy's PostFilter sends) take the same walk for one subject's pairs where no Check
 * of the permission can end at the dispatch-depth limit -- by the schema (no recursion), or, for a recursive permission, on the snapshot at hand: one forward
 * sweep over the type's objects for a subject nobody is decides that for every subject (acl_stats_t.depth_sweeps; the first call at a snapshot goes forward,
 * the second sweeps; ACL_DEPTH_SWEEP=0 in the environment switches the sweep off).  Where the sweep FINDS such objects (a cycle of groups behind some resources)
 * their pairs answer ACL_ERR_DEPTH as the forward path does, out of the sweep's bitmap. */
int acl_check_bulk_keep_v(acl_engine_t *h, const acl_check_item_v_t *items, size_t n, const uint32_t *item_off, size_t k_items, uint8_t *keep_out);
int acl_check_bulk_keep_packed(acl_engine_t *h, const acl_packed_request_t *req, const uint32_t *item_off, size_t k_items, uint8_t *keep_out);
int acl_check_bulk_keep_ids_device(acl_engine_t *h, const void *d_items, size_t n, const void *d_item_off, size_t k_items, void *d_keep_out);
/* PostFilter at LIST level: filterListResponse (postfilter.go:17-55).  body = the kube list response (JSON).  Every element
 * of its "items" array gets one check per template, rendered from the item's metadata: placeholders {{name}}, {{namespace}},
 * {{namespacedName}} (namespace/name, or name when cluster scoped) and {{user.name}} inside `type:id#perm@type:id[#rel]`
 * (the form of deploy/rules.yaml:68; richer rule expressions resolve in Go and use acl_check_bulk_keep).  A template that
 * does not resolve for an item contributes no check (postfilter.go:92-95); an item is kept iff all its checks are
 * HAS_PERMISSION (postfilter.go:144-178).  *out_body (release with acl_free) is the ORIGINAL document with the dropped
 * items' bytes cut out ("items": null when none is left, as the reference's nil slice marshals); a body without an
 * "items" array, or with an empty one, comes back unchanged (postfilter.go:26-35).  Invalid JSON: ACL_ERR_INVALID_ARGUMENT. */
int acl_filter_list_response(acl_engine_t *h, const char *body, size_t body_len, const char *const *templates, size_t n_templates,
                             const char *user_name, char **out_body, size_t *out_len, uint64_t *kept_out, uint64_t *total_out);
/* ... with the kube request the list answers: every item's placeholders are resolved as rules.NewResolveInput does (pkg/rules/rules.go:315-342) --
 * the item's own metadata.name / metadata.namespace first, the request's where the item has none, and no namespace at all when the request's
 * resource is `namespaces`.  NULL req (or NULL members) = the item's metadata only. */
typedef struct {
    const char *name;       /* RequestInfo.Name */
    const char *namespace_; /* RequestInfo.Namespace */
    const char *resource;   /* RequestInfo.Resource, e.g. "pods", "namespaces" */
} acl_list_request_t;
int acl_filter_list_response_req(acl_engine_t *h, const char *body, size_t body_len, const char *const *templates, size_t n_templates,
                                 const char *user_name, const acl_list_request_t *req, char **out_body, size_t *out_len, uint64_t *kept_out,
                                 uint64_t *total_out);
/* PreFilter: prefilterResult.IsAllowed (lookups.go:25-36; consumers responsefilterer.go:349-415) over the bitmap of
 * acl_lookup_resources*: allowed_out[i] = 1 iff object_ids[i] (the rule's `ns/name` object id text) is set. */
int acl_bitmap_test_names(acl_engine_t *h, int type, const uint32_t *bitmap, size_t bitmap_words, const char *const *object_ids, size_t n,
                          uint8_t *allowed_out);
/* PreFilter consumers over the kube response's BYTES (responsefilterer.go:349-372 filterTable, :374-399 filterList, :401-416 filterObject):
 * keeps the elements prefilterResult.IsAllowed(namespace, name) admits.  The reference maps every LookupResources id to a NamespacedName with
 * the rule's fromObjectIDNameExpr / fromObjectIDNamespaceExpr (lookups.go:98-131) and tests set membership; here `id_template` is that mapping
 * read the other way -- the object id of an item's (namespace, name), in the placeholder form of acl_filter_list_response: "{{name}}" for
 * deploy/rules.yaml:51 (`{{resourceId}}`), "{{namespacedName}}" for rules.yaml:105-106 (split_namespace / split_name) -- and the id's bit in the
 * bitmap of acl_lookup_resources* decides.  ACL_BODY_LIST: "items" (each item's metadata); ACL_BODY_TABLE: "rows" (each row's
 * object.metadata; a row without a decodable object: ACL_ERR_INVALID_ARGUMENT, as the reference's decode error); an array without survivors
 * becomes [] (both consumers start from an empty, non-nil slice).  ACL_BODY_OBJECT: the body comes back unchanged when its object is allowed,
 * else ACL_ERR_PERMISSION_DENIED ("unauthorized": writeResp makes a 401 of it, responsefilterer.go:716-727).  The answer is the original
 * bytes minus the dropped elements (release with acl_free); a body without the array comes back unchanged.  Needs no GPU.  (split_name /
 * split_namespace cut at the first '/', pkg/rules/env.go:18-56: the one id shape the forward reading cannot reach is an id that BEGINS with '/'.) */
typedef enum { ACL_BODY_LIST = 0, ACL_BODY_TABLE = 1, ACL_BODY_OBJECT = 2 } acl_body_kind_t;
int acl_prefilter_response(acl_engine_t *h, int type, const uint32_t *bitmap, size_t bitmap_words, const char *id_template, int kind, const char *body,
                           size_t body_len, char **out_body, size_t *out_len, uint64_t *kept_out, uint64_t *total_out);
/* Watch (watch.go:29-38): every update committed by acl_write / acl_delete_by_filter with revision > after_revision
 * whose resource type is in `types` (ntypes == 0: all), in commit order; op = ACL_OP_TOUCH or ACL_OP_DELETE.
 * after_revision == UINT64_MAX or cb == NULL only reports the head revision ("start from now").  *revision_out
 * is the cursor for the next poll.  ACL_ERR_OUT_OF_RANGE when the cursor fell out of the retained feed. */
typedef void (*acl_watch_cb)(void *user, uint64_t revision, int32_t op, const acl_relationship_t *rel);
int acl_watch_poll(acl_engine_t *h, uint64_t after_revision, const int *types, int ntypes, acl_watch_cb cb, void *user, uint64_t *revision_out);
/* The blocking half of the stream (watch.go:38 blocks in Recv()): returns ACL_OK as soon as the feed holds an update with revision >
 * after_revision whose resource type is in `types` (then poll), else ACL_ERR_DEADLINE_EXCEEDED / ACL_ERR_CANCELLED by `opts` (NULL: waits
 * for ever).  A condition variable behind acl_write / acl_delete_by_filter: no sleeping poll per open watch. */
int acl_watch_wait(acl_engine_t *h, uint64_t after_revision, const int *types, int ntypes, const acl_call_opts_t *opts, uint64_t *revision_out);
/* RunWatch's loop body for a whole poll (watch.go:38-108): every update of templ->resource_type behind the cursor and, for all of them, ONE
 * bulk Check of `templ.resource_type : <the update's resource id> # templ.permission @ templ's subject` (the reference issues one
 * CheckPermission per update, watch.go:50-67); cb once per update, in commit order, with the decision (permissionship, per-item error). */
typedef void (*acl_watch_check_cb)(void *user, uint64_t revision, int32_t op, const acl_relationship_t *rel, uint8_t permissionship, int32_t err);
int acl_watch_recheck(acl_engine_t *h, uint64_t after_revision, const acl_check_item_t *templ, acl_watch_check_cb cb, void *user, uint64_t *revision_out);
/* Micro-batching front-end for the proxy's call shape -- many concurrent 1-item checks (check.go:76-94 one goroutine
 * per check expression, watch.go:50 one per update).  acl_check_one blocks its caller; while a batcher runs,
 * concurrent callers share ONE device pass (drained after at most max_wait_us or when max_items are waiting). */
int acl_batcher_start(acl_engine_t *h, uint32_t max_items, uint32_t max_wait_us);
int acl_batcher_stop(acl_engine_t *h);
int acl_batcher_stats(acl_engine_t *h, uint64_t *batches, uint64_t *items);
int acl_check_one(acl_engine_t *h, const acl_check_item_t *item, uint8_t *perm_out, int32_t *err_out);
int acl_check_one_opts(acl_engine_t *h, const acl_check_item_t *item, uint8_t *perm_out, int32_t *err_out, const acl_call_opts_t *opts);
/* The same check without an OS thread blocked per request -- the form a cgo shim should bind: the goroutine that issues
 * check.go:48 / watch.go:50 parks on a Go channel and ONE poller goroutine drains the completions.  With
 * acl_check_one every check costs the host a futex sleep + wake-up (~17 us of kernel time per check on the measured hosts:
 * a ~0.9 M checks/s ceiling on 16 cores, whatever the device does); here a wake-up is paid per device pass.
 * acl_check_one_submit returns at once (needs a running batcher; `tag` is the caller's); acl_check_completions takes up to
 * `max` finished checks, blocking while there are none (timeout_ns < 0: until one arrives, 0: never, > 0: at most that
 * long).  rc / err / perm are what acl_check_one would have returned / stored.  Any number of threads may submit and poll;
 * each completion is delivered once; completions not yet collected when the batcher stops stay collectable. */
typedef struct {
    uint64_t tag;
    int32_t rc;   /* ACL_OK, or the status of the device pass that carried the item */
    int32_t err;  /* per-item error (0 = none), as acl_check_one's err_out */
    uint8_t perm; /* ACL_PERM_* */
    uint8_t pad[3];
} acl_completion_t;
int acl_check_one_submit(acl_engine_t *h, const acl_check_item_t *item, uint64_t tag);
int acl_check_completions(acl_engine_t *h, acl_completion_t *out, size_t max, int64_t timeout_ns, size_t *n_out);
/* LookupResources in the same form (responsefilterer.go:165-204: every prefilter runs in a goroutine of its own next to the upstream call):
 * submit returns at once, the answer arrives tagged through acl_lookup_completions.  `bitmap` is an engine-allocated row over the result type's
 * ids (`words` 32-bit words, sized when the walk runs -- objects created after the submit are covered); the receiver releases it with acl_free.
 * rc != 0: the lookup failed (bitmap NULL).  Concurrent submissions of one (resource type, permission, subject class) share one reverse walk. */
typedef struct {
    uint64_t tag;
    int32_t rc;
    uint32_t reserved;
    uint64_t count; /* ids in the row */
    size_t words;
    uint32_t *bitmap;
} acl_lookup_completion_t;
int acl_lookup_one_submit(acl_engine_t *h, const char *rtype, const char *perm, const char *stype, const char *sid, const char *srel, uint64_t tag);
int acl_lookup_completions(acl_engine_t *h, acl_lookup_completion_t *out, size_t max, int64_t timeout_ns, size_t *n_out);
/* the same for Filter requests (lookups.go:65; one LookupResources per list request, each from its own goroutine:
 * responsefilterer.go:165): concurrent requests with the same (resource type, permission, subject class) share ONE
 * batched reverse walk.  Arguments as acl_lookup_resources. */
int acl_lookup_one(acl_engine_t *h, const char *resource_type, const char *permission, const char *subject_type, const char *subject_id,
                   const char *subject_relation, uint32_t *bitmap_out, size_t bitmap_words, uint64_t *count_out);
int acl_lookup_one_opts(acl_engine_t *h, const char *resource_type, const char *permission, const char *subject_type, const char *subject_id,
                        const char *subject_relation, uint32_t *bitmap_out, size_t bitmap_words, uint64_t *count_out, const acl_call_opts_t *opts);
int acl_batcher_lookup_stats(acl_engine_t *h, uint64_t *walks, uint64_t *lookups);

/* ---- sharded graph: the north star's multi-GPU configuration ----
 * One engine per GPU holds the rows of the object types with hash(type name) mod world == rank (FNV-1a, avalanched: acl_shard_of_type); the
 * relationship store (ids, writes) stays replicated.  A batch advances one dispatch level at a time on every
 * shard; sub-checks whose rows live elsewhere are appended to the caller's export buffer (16-byte frontier
 * entries) and the host exchanges them between levels -- RCCL all-gather in aclgpu/sharded.py, ncclAllGather in
 * the cgo shim -- then hands every other shard's segment to *_import.  The non-sharded evaluating entry points
 * fail with ACL_ERR_FAILED_PRECONDITION on a sharded engine.  All buffers are device pointers owned by the caller. */
typedef struct {
    uint64_t exported; /* entries this step wanted to export; > export_cap means entries were dropped: redo the batch with a larger buffer */
    uint32_t produced; /* != 0: this shard's own next frontier is not empty */
    uint32_t overflow; /* 0 ok; 1 frontier out of chunks (acl_shard_grow_frontier, redo the batch); 2 a row exceeds the enumeration limit */
} acl_shard_step_t;
enum { ACL_SHARD_VISIT = 1, ACL_SHARD_EXPAND = 2 };
int acl_shard_configure(acl_engine_t *h, uint32_t rank, uint32_t world); /* world == 1 restores the single-GPU engine */
int acl_shard_of_type(acl_engine_t *h, int type);                        /* shard holding the type's rows; -1 if unknown */
int acl_shard_grow_frontier(acl_engine_t *h);
/* hipStream_t every acl_shard_* call launches on (the protocol's own evaluation context): the host orders its
 * collectives after / before the steps on this stream.  NULL without a GPU. */
void *acl_shard_stream(acl_engine_t *h);
/* Check (check.go:48): d_items = the WHOLE batch on every shard (each seeds the items whose resource type it owns);
 * d_has / d_err = n bytes each, per shard; after the last level the host MAX-reduces both across shards. */
int acl_shard_check_begin(acl_engine_t *h, const void *d_items, size_t n, void *d_has, void *d_err);
int acl_shard_check_step(acl_engine_t *h, uint32_t level /* 1..50 */, void *d_has, void *d_err, void *d_export, size_t export_cap,
                         acl_shard_step_t *out);
/* all-to-all form: d_export is `world` buffers of cap_per_dest entries, buffer d holds the entries owned by shard d;
 * exported_by_dest[d] counts them (out->exported = the largest).  The host sends buffer d to rank d and imports what it
 * receives; G times fewer bytes than the all-gather form. */
int acl_shard_check_step_by_dest(acl_engine_t *h, uint32_t level, void *d_has, void *d_err, void *d_export, size_t cap_per_dest,
                                 acl_shard_step_t *out, uint64_t *exported_by_dest /* [world] */);
int acl_shard_check_import(acl_engine_t *h, uint32_t level, const void *d_entries, size_t n);
int acl_shard_check_finish(acl_engine_t *h, const void *d_has, const void *d_err, size_t n, void *d_perm_out, void *d_err_out);
/* LookupResources (lookups.go:65) for n subjects of one class.  Iteration 1 expands the seeds (ACL_SHARD_EXPAND);
 * then ACL_SHARD_VISIT (marks first visits, exports the states other shards hold parent rows for), exchange +
 * import, ACL_SHARD_EXPAND, ... until no shard produced or exported anything in a visit step. */
int acl_shard_lookup_begin(acl_engine_t *h, int rtype, int permission, int stype, int srel /* -1 none */, const uint32_t *subject_ids, size_t n);
int acl_shard_lookup_step(acl_engine_t *h, uint32_t iter, int phase, void *d_export, size_t export_cap, acl_shard_step_t *out);
int acl_shard_lookup_import(acl_engine_t *h, uint32_t iter, const void *d_entries, size_t n);
/* n rows of bitmap_words words: the resource type's owner returns the result bits, every other shard zeros */
int acl_shard_lookup_finish(acl_engine_t *h, void *d_bitmaps_out, size_t bitmap_words);

/* ---- native sharded loop: the whole level loop of a sharded Check / LookupResources inside the library ----
 * SPMD: every shard makes the same call with the same batch.  Per level the shards exchange 16-byte HEADERS (entries exported, produced flag,
 * overflow code: written on the device) and -- on the levels the previous batch exported something, or on all of them -- fixed-capacity blocks
 * of frontier ENTRIES; imports, termination and retry decisions are computed on the device from the headers, identically on every shard; the
 * host synchronises once per burst of levels.  A level that exports where none was expected makes every shard redo the batch with entries on
 * every level.  Check results (has / err) and LookupResources rows are MAX-reduced across shards once per batch.  The collective is a set of
 * callbacks: acl_shard_rccl_* below supply RCCL (ncclAllGather / ncclAllReduce, grouped ncclSend / ncclRecv over xGMI); a host with its own
 * ncclComm_t (the cgo shim) or a test double plugs in the same way.  Callbacks enqueue on `hip_stream` and return 0 or an ACL_ERR_* code.
 * all_to_all may be NULL: Check then exchanges one block per shard through all_gather (every shard receives everything and keeps what it
 * owns) instead of one block per (shard, destination) -- `world` times the bytes.  LookupResources travels the same way: a visited state
 * goes into the block of every shard that holds parent rows for its slot (all_to_all), or to everybody (all_gather).
 * Schemas with intersection / exclusion / `.all()` (reference pkg/spicedb/spicedb.go:19-24 boots any schema): evaluated by these two entry points
 * (the host-driven step protocol above refuses them) -- leaf cells come from per-shard ranges of one global cell space, all_reduce_max_u8 then
 * covers the batch's cells as well as its answers and the shards' combine nodes travel through all_gather once per batch. */
typedef struct {
    void *user;
    int (*all_gather)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, void *hip_stream);
    int (*all_reduce_max_u8)(void *user, void *d_buf, size_t n_bytes, void *hip_stream); /* in place */
    int (*all_to_all)(void *user, const void *d_send, void *d_recv, size_t bytes_per_peer, void *hip_stream); /* block r of send -> rank r; block r of recv <- rank r */
} acl_shard_comm_t;
typedef struct {
    uint32_t levels;            /* dispatch levels the batch needed (max over shards) */
    uint32_t exchanges;         /* header exchanges issued (includes the burst's levels past the end) */
    uint32_t host_syncs;        /* stream synchronisations: bursts + the final one */
    uint32_t retries;           /* batch redone after an export block / frontier overflow, or an export on a level planned without entries */
    uint64_t exchanged_bytes;   /* bytes this shard received in the exchanges */
    uint64_t entries_exchanged; /* frontier entries exported by all shards */
    uint64_t export_capacity;   /* entries per exchange block at the end */
    uint32_t data_exchanges;    /* ... of which also moved entry blocks */
    uint32_t reserved;
} acl_shard_bulk_stats_t;
int acl_shard_check_bulk(acl_engine_t *h, const acl_shard_comm_t *comm, const void *d_items, size_t n, void *d_perm_out, void *d_err_out,
                         acl_shard_bulk_stats_t *stats_out);
/* LookupResources for n subjects (host array) of one class: n rows of bitmap_words words in device memory, the same on every shard afterwards */
int acl_shard_lookup_bulk(acl_engine_t *h, const acl_shard_comm_t *comm, int rtype, int permission, int stype, int srel /* -1 none */, const uint32_t *subject_ids,
                          size_t n, void *d_bitmaps_out, size_t bitmap_words, acl_shard_bulk_stats_t *stats_out);
#define ACL_RCCL_UNIQUE_ID_BYTES 128
int acl_shard_rccl_unique_id(void *id_out /* ACL_RCCL_UNIQUE_ID_BYTES; rank 0 makes it, the host hands it to every rank */);
int acl_shard_rccl_init(acl_engine_t *h, const void *unique_id, uint32_t rank, uint32_t world); /* also acl_shard_configure(rank, world) */
int acl_shard_rccl_destroy(acl_engine_t *h);
int acl_shard_check_bulk_rccl(acl_engine_t *h, const void *d_items, size_t n, void *d_perm_out, void *d_err_out, acl_shard_bulk_stats_t *stats_out);
int acl_shard_lookup_bulk_rccl(acl_engine_t *h, int rtype, int permission, int stype, int srel, const uint32_t *subject_ids, size_t n, void *d_bitmaps_out,
                               size_t bitmap_words, acl_shard_bulk_stats_t *stats_out);

/* ---- test hook ----
 * Updates the HOST copy of the snapshot the way the next read would (in-place patch from the change feed when
 * possible, rebuild otherwise) and verifies it against the relationship store: every live relationship findable,
 * nothing dead left, rows sorted, no unsound leaf flag.  Store-only engines only (it never touches a device). */
int acl_selfcheck_snapshot(acl_engine_t *h, int *patched_out); /* *patched_out: 1 patched in place, 0 rebuilt, 2 was current */
/* The host half of the background snapshot compaction in two steps (store-only engines): phase 0 builds a snapshot from a
 * copy-on-write view of the store, phase 1 catches it up with the writes since (the ordinary patcher), adopts and verifies it.
 * *adopted_out = 0 when the catch-up was not expressible as a patch (the engine then rebuilds synchronously). */
int acl_selfcheck_compaction(acl_engine_t *h, int phase, int *adopted_out);
/* The elements of the JSON array at body[arr_open] (`[`) as the list-level filters find them (csrc/engine_list.cpp scan_array): chunk_bytes > 0 forces the
 * parallel index (csrc/json_index.hpp) with chunks of that size, 0 = what a call of that size would do.  spans_out: {begin, end} byte offsets per element
 * (cap elements at most), *n_out elements, *close_out the offset of `]`.  ACL_ERR_INVALID_ARGUMENT when the array or one of its elements is not JSON. */
int acl_selfcheck_json_array(acl_engine_t *h, const char *body, size_t body_len, size_t arr_open, size_t chunk_bytes, size_t *spans_out, size_t cap, size_t *n_out,
                             size_t *close_out);
/* ---- measurement ---- */
typedef struct {
    uint64_t check_items;      /* items answered since open / last reset */
    uint64_t check_passes;     /* device passes (sub-batches) */
    uint64_t expand_launches;  /* frontier-expansion kernel launches */
    uint64_t levels_last;      /* levels the last pass needed */
    uint64_t frontier_entries; /* frontier entries produced (all levels) */
    double kernel_ms;          /* HIP-event time of all engine kernels since reset (needs timing on) */
    double expand_ms;          /* ... of the frontier-expansion kernel only */
    uint64_t snapshot_edges;   /* edges in the current HBM snapshot */
    uint64_t snapshot_bytes;   /* bytes of the current HBM snapshot */
    uint64_t snapshot_builds;
    uint64_t overflow_retries;
    uint64_t snapshot_edges_local; /* relationships whose rows THIS engine holds (== snapshot_edges unless sharded) */
    uint64_t snapshot_patches;     /* times committed writes were patched into the HBM snapshot in place (no rebuild) */
    double local_ms;               /* HIP-event time of the single-launch small-batch kernel (k_check_local) */
    uint64_t local_passes;         /* device passes answered by that kernel (one launch for all levels) */
    uint64_t snapshot_compactions; /* snapshots rebuilt in the background and swapped in */
    double rev_local_ms;           /* HIP-event time of the single-launch LookupResources kernel (k_rev_local) */
    uint64_t rev_local_passes;     /* LookupResources groups answered by that kernel (one launch for all reverse levels) */
    uint64_t lookup_requests;      /* LookupResources requests answered since open / last reset */
    uint64_t ids_recycled;         /* object ids given a new name after their object had lost its last relationship (since the schema was loaded) */
    uint64_t keep_route_calls;     /* acl_check_bulk_keep_v / _packed calls answered by ONE reverse walk and bit tests (since open) */
    uint64_t depth_sweeps;         /* forward sweeps over a whole type that established "no Check of this permission ends at the depth limit" for the snapshot (since open) */
} acl_stats_t;
int acl_stats(acl_engine_t *h, acl_stats_t *out);
int acl_stats_reset(acl_engine_t *h);
int acl_set_timing(acl_engine_t *h, int on); /* bracket every kernel with HIP events on its context's stream */

#ifdef __cplusplus
}
#endif
#endif /* ACLGPU_H */
