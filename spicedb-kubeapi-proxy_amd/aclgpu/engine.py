"""Engine: thin Python handle over the C ABI (include/aclgpu.h).

Numeric entry points take / return numpy arrays; the string entry points mirror
the request shapes the reference builds in pkg/authz/check.go:23-39 and
pkg/authz/lookups.go:49-62.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import CallOpts, CheckItem, Completion, Config, Filter, READ_CB, Relationship, Stats, Update, WATCH_CB, WATCH_CHECK_CB

PERM_UNSPECIFIED, PERM_NO, PERM_HAS, PERM_CONDITIONAL = 0, 1, 2, 3
OP_CREATE, OP_TOUCH, OP_DELETE = 1, 2, 3
PRE_MUST_NOT_MATCH, PRE_MUST_MATCH = 1, 2
ERR_INVALID_ARGUMENT, ERR_NOT_FOUND, ERR_ALREADY_EXISTS, ERR_RESOURCE_EXHAUSTED, ERR_FAILED_PRECONDITION = 3, 5, 6, 8, 9
ERR_INTERNAL, ERR_UNAVAILABLE, ERR_DEPTH = 13, 14, 100
ERR_OUT_OF_RANGE = 11
ERR_CANCELLED, ERR_DEADLINE_EXCEEDED = 1, 4
WATCH_FROM_NOW = 0xFFFFFFFFFFFFFFFF
NO_RELATION = 0xFFFF

ITEM_DTYPE = np.dtype([("resource_type", "<u2"), ("permission", "<u2"), ("resource_id", "<u4"), ("subject_type", "<u2"),
                       ("subject_relation", "<u2"), ("subject_id", "<u4")])
assert ITEM_DTYPE.itemsize == 16


class AclError(Exception):
    """Carries the gRPC status code the Go shim would return (status.Code(err))."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.message = msg


def _b(s):
    return None if s is None else s.encode()


class Engine:
    names_invalid_fields = True  # a request the API's validation refuses fails with a message naming the item, the field, the value and the pattern (tests/kat_runner.py)

    def __init__(self, schema: str | None = None, relationships: str | None = None, device: int = -1, frontier_entries: int = 0,
                 max_sub_batch: int = 0, store_only: bool = False, contexts: int = 0, devices=None, per_item_validation: bool = False,
                 lenient_lookup: bool = False, eager_contexts: bool = False):
        """devices: HIP ordinals of the replicas (acl_open_replicas: ONE store and one set of name tables in front of one HBM snapshot per
        entry; a device may be listed more than once); default one replica on `device`."""
        self._L = _lib.load()
        cfg = Config(device, frontier_entries, max_sub_batch, (1 if store_only else 0) | (2 if per_item_validation else 0) | (4 if lenient_lookup else 0) | (8 if eager_contexts else 0), contexts, 0)
        # (ACL_FLAG_STORE_ONLY | _PER_ITEM_VALIDATION | _LENIENT_LOOKUP | _EAGER_CONTEXTS)
        h = C.c_void_p()
        if devices:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = self._L.acl_open_replicas(C.byref(cfg), arr, len(devices), C.byref(h))
        else:
            rc = self._L.acl_open(C.byref(cfg), C.byref(h))
        self._h = h if rc == 0 else None
        self._check(rc)
        if schema is not None:
            self.load_bootstrap(schema, relationships)

    # ---- plumbing
    def _check(self, rc):
        if rc:
            raise AclError(rc, (self._L.acl_last_error() or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.acl_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def load_bootstrap(self, schema: str, relationships: str | None = None):
        s = schema.encode()
        r = relationships.encode() if relationships else None
        self._check(self._L.acl_load_bootstrap(self._h, s, len(s), r, len(r) if r else 0))

    def load_bootstrap_yaml(self, yaml_text):
        """The bootstrap FILE form (pkg/spicedb/bootstrap.yaml; spicedb.go:19-24): YAML `{schema, relationships}` documents."""
        y = yaml_text if isinstance(yaml_text, bytes) else yaml_text.encode()
        self._check(self._L.acl_load_bootstrap_yaml(self._h, y, len(y)))

    # ---- ids
    def type_id(self, t: str) -> int:
        return self._L.acl_type_id(self._h, _b(t))

    def relation_id(self, t: str, r: str | None) -> int:
        return -1 if not r else self._L.acl_relation_id(self._h, self.type_id(t), _b(r))

    def intern(self, t: str, oid: str) -> int:
        out = C.c_uint32()
        self._check(self._L.acl_intern(self._h, self.type_id(t), _b(oid), C.byref(out)))
        return out.value

    def find(self, t: str, oid: str):
        out = C.c_uint32()
        rc = self._L.acl_find(self._h, self.type_id(t), _b(oid), C.byref(out))
        return out.value if rc == 0 else None

    def object_name(self, t: str, i: int):
        """acl_object_name_copy: the name is copied out under the names lock (an id that takes part in no relationship may be renamed later)"""
        buf = C.create_string_buffer(256)
        while True:  # (the id may be given a longer name between two copies: a length is only used once it fits the buffer it was copied into)
            n = self._L.acl_object_name_copy(self._h, self.type_id(t), int(i), buf, len(buf))
            if n < 0:
                return None
            if n < len(buf):
                return buf.raw[:n].decode()
            buf = C.create_string_buffer(n + 1)

    def object_count(self, t: str) -> int:
        return self._L.acl_object_count(self._h, self.type_id(t))

    # ---- writes
    @staticmethod
    def _mkrel(r, expires=0):
        rt, rid, rel, st, sid, srel = r
        return Relationship(_b(rt), _b(rid), _b(rel), _b(st), _b(sid), _b(srel or ""), int(expires))

    @staticmethod
    def _mkfilter(op=0, rtype=None, rid=None, rel=None, stype=None, sid=None, srel=None):
        return Filter(op, _b(rtype), _b(rid), _b(rel), _b(stype), _b(sid), _b(srel))

    def write(self, updates, preconditions=()):
        """updates: [(op, (rt, rid, rel, st, sid, srel) | 'text', expires?)]; preconditions: [(op, {filter kwargs})]."""
        from .text import parse_relationship
        ups = (Update * max(1, len(updates)))()
        for i, u in enumerate(updates):
            r = parse_relationship(u[1]) if isinstance(u[1], str) else u[1]
            ups[i] = Update(u[0], self._mkrel(r, u[2] if len(u) > 2 else 0))
        pre = (Filter * max(1, len(preconditions)))()
        for i, (op, f) in enumerate(preconditions):
            pre[i] = self._mkfilter(op, **f)
        rev = C.c_uint64()
        self._check(self._L.acl_write(self._h, ups, len(updates), pre, len(preconditions), C.byref(rev)))
        return rev.value

    def touch(self, *rels):
        return self.write([(OP_TOUCH, r) for r in rels])

    def delete_by_filter(self, preconditions=(), **f):
        """DeleteRelationships; preconditions [(op, {filter kwargs})] are evaluated against the pre-delete state."""
        n, rev = C.c_uint64(), C.c_uint64()
        flt = self._mkfilter(0, **f)
        pre = (Filter * max(1, len(preconditions)))()
        for i, (op, pf) in enumerate(preconditions):
            pre[i] = self._mkfilter(op, **pf)
        self._check(self._L.acl_delete_by_filter_pre(self._h, C.byref(flt), pre, len(preconditions), C.byref(n), C.byref(rev)))
        return n.value

    def read(self, **f):
        out = []

        def cb(_u, rp):
            r = rp.contents
            out.append((r.resource_type.decode(), r.resource_id.decode(), r.relation.decode(), r.subject_type.decode(), r.subject_id.decode(),
                        (r.subject_relation or b"").decode(), r.expires_at))

        flt = self._mkfilter(0, **f)
        self._check(self._L.acl_read(self._h, C.byref(flt), READ_CB(cb), None))
        return out

    def add_edges(self, rtype, rel, stype, srel, res, subj):
        res = np.ascontiguousarray(res, dtype=np.uint32)
        subj = np.ascontiguousarray(subj, dtype=np.uint32)
        assert res.shape == subj.shape
        # srel "*": `stype:*` relationships, one per entry of res (subj is ignored)
        self._check(self._L.acl_add_edges(self._h, self.type_id(rtype), self.relation_id(rtype, rel), self.type_id(stype),
                                          -2 if srel == "*" else self.relation_id(stype, srel), res.size, res.ctypes.data, subj.ctypes.data))

    def set_now(self, t: int):
        self._check(self._L.acl_set_now(self._h, int(t)))

    @property
    def revision(self):
        return self._L.acl_revision(self._h)

    def snapshot(self):
        self._check(self._L.acl_snapshot(self._h))

    # ---- checks
    def check_bulk(self, items):
        """items: [(rt, rid, perm, st, sid, srel)] -> (perms list, errs list); index aligned (check.go:54-57)."""
        n = len(items)
        arr = (CheckItem * max(1, n))()
        for i, it in enumerate(items):
            arr[i] = CheckItem(*[_b(x if x is not None else "") for x in it])
        perm = np.zeros(max(1, n), dtype=np.uint8)
        err = np.zeros(max(1, n), dtype=np.int32)
        self._check(self._L.acl_check_bulk(self._h, arr, n, perm.ctypes.data, err.ctypes.data))
        return perm[:n].tolist(), err[:n].tolist()

    def check(self, rt, rid, perm, st, sid, srel=""):
        """CheckPermission: (permissionship, error code).  A request the API's validation refuses fails as a whole (validate.hpp): for a
        single check the call's InvalidArgument is the item's."""
        try:
            p, e = self.check_bulk([(rt, rid, perm, st, sid, srel)])
        except AclError as x:
            if x.code == ERR_INVALID_ARGUMENT:
                return 0, x.code
            raise
        return p[0], e[0]

    def make_items(self, rtype, perm, res, stype, srel, subj):
        """Interned 16-byte items (acl_item_t) for one (type#perm, subject class) and id arrays."""
        res = np.asarray(res, dtype=np.uint32)
        items = np.zeros(res.size, dtype=ITEM_DTYPE)
        items["resource_type"] = self.type_id(rtype)
        items["permission"] = self.relation_id(rtype, perm)
        items["resource_id"] = res
        items["subject_type"] = self.type_id(stype)
        r = self.relation_id(stype, srel)
        items["subject_relation"] = NO_RELATION if r < 0 else r
        items["subject_id"] = np.asarray(subj, dtype=np.uint32)
        return items

    def check_bulk_ids(self, items: np.ndarray):
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n = items.size
        perm = np.zeros(max(1, n), dtype=np.uint8)
        err = np.zeros(max(1, n), dtype=np.int32)
        self._check(self._L.acl_check_bulk_ids(self._h, items.ctypes.data, n, perm.ctypes.data, err.ctypes.data))
        return perm[:n], err[:n]

    def check_bulk_ids_into(self, items: np.ndarray, perm: np.ndarray, err: np.ndarray):
        """acl_check_bulk_ids into caller-owned arrays (e.g. views of host_alloc memory: no staging copy)."""
        self._check(self._L.acl_check_bulk_ids(self._h, items.ctypes.data, items.size, perm.ctypes.data, err.ctypes.data))

    def make_check_strings(self, rtype, perm, res, stype, subj, srel=""):
        """A prepared acl_check_item_t array whose object ids are the DECIMAL numeric ids (bench.py's string leg)."""
        n = len(res)
        arr = (CheckItem * max(1, n))()
        rt, pm, st, sr = _b(rtype), _b(perm), _b(stype), _b(srel)
        keep = []
        for i in range(n):
            a, b = str(int(res[i])).encode(), str(int(subj[i])).encode()
            keep.append(a)
            keep.append(b)
            arr[i] = CheckItem(rt, a, pm, st, b, sr)
        return arr, n, keep

    def make_check_strings_named(self, items):
        """A prepared acl_check_item_t array of [(rt, rid, perm, st, sid, srel)] (the ctypes marshalling done once, outside any timing)."""
        n = len(items)
        arr = (CheckItem * max(1, n))()
        keep = []
        for i, it in enumerate(items):
            bs = [_b(x if x is not None else "") for x in it]
            keep.append(bs)
            arr[i] = CheckItem(*bs)
        return arr, n, keep

    def make_check_views(self, items):
        """A prepared acl_check_item_v_t array ({pointer, length} x 6 per item) of [(rt, rid, perm, st, sid, srel)]: every distinct string is
        stored once in one blob WITHOUT a NUL behind it, and equal strings share their bytes -- as a cgo shim pointing at Go strings would."""
        n = len(items)
        blob = bytearray()
        where = {}
        views = np.zeros((max(1, n), 6, 2), dtype=np.uint64)
        offs = np.zeros((max(1, n), 6), dtype=np.int64)
        for i, it in enumerate(items):
            for f, x in enumerate(it):
                b = _b(x if x is not None else "") or b""
                if f == 5 and not b:
                    offs[i, f] = -1  # an absent subject relation is {NULL, 0}
                    continue
                at = where.get(b)
                if at is None:
                    at = where[b] = len(blob)
                    blob += b
                offs[i, f] = at
                views[i, f, 1] = len(b)
        buf = np.frombuffer(bytes(blob) or b"\0", dtype=np.uint8).copy()
        views[:, :, 0] = np.where(offs >= 0, buf.ctypes.data + offs, 0).astype(np.uint64)
        return views, n, buf

    def check_bulk_views(self, prepared, cancel=None, timeout_s=None):
        """acl_check_bulk_v; with `cancel` (a ctypes c_int32 the caller raises) or `timeout_s`: acl_check_bulk_v_opts -- CheckBulkPermissions(ctx, ...)"""
        views, n, _blob = prepared
        perm = np.zeros(max(1, n), dtype=np.uint8)
        err = np.zeros(max(1, n), dtype=np.int32)
        if cancel is None and timeout_s is None:
            self._check(self._L.acl_check_bulk_v(self._h, views.ctypes.data, n, perm.ctypes.data, err.ctypes.data))
        else:
            o = _lib.CallOpts(C.pointer(cancel) if cancel is not None else None, int((timeout_s or 0) * 1e9))
            self._check(self._L.acl_check_bulk_v_opts(self._h, views.ctypes.data, n, perm.ctypes.data, err.ctypes.data, C.byref(o)))
        return perm[:n], err[:n]

    def resolve_bulk_views(self, prepared):
        """acl_resolve_bulk_v: the prepared views as the 16-byte items of the id entry points (no device pass) + a per-item error array."""
        views, n, _blob = prepared
        items = np.zeros(max(1, n), dtype=ITEM_DTYPE)
        err = np.zeros(max(1, n), dtype=np.int32)
        self._check(self._L.acl_resolve_bulk_v(self._h, views.ctypes.data, n, items.ctypes.data, err.ctypes.data))
        return items[:n], err[:n]

    def check_bulk_prepared(self, prepared):
        arr, n, _keep = prepared
        perm = np.zeros(max(1, n), dtype=np.uint8)
        err = np.zeros(max(1, n), dtype=np.int32)
        self._check(self._L.acl_check_bulk(self._h, arr, n, perm.ctypes.data, err.ctypes.data))
        return perm[:n], err[:n]

    @staticmethod
    def _opts(cancel=None, timeout_s=None):
        """cancel: a ctypes.c_int32 the caller sets non-zero to abandon the call (the C side of ctx.Done())."""
        if cancel is None and not timeout_s:
            return None
        return CallOpts(C.pointer(cancel) if cancel is not None else None, int((timeout_s or 0) * 1e9))

    def check_bulk_ids_opts(self, items: np.ndarray, cancel=None, timeout_s=None):
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        n = items.size
        perm = np.zeros(max(1, n), dtype=np.uint8)
        err = np.zeros(max(1, n), dtype=np.int32)
        o = self._opts(cancel, timeout_s)
        self._check(self._L.acl_check_bulk_ids_opts(self._h, items.ctypes.data, n, perm.ctypes.data, err.ctypes.data, C.byref(o) if o else None))
        return perm[:n], err[:n]

    # ---- pipelined host-buffer path: pinned request / answer arrays + submit / wait
    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Page-locked host memory as a uint8 array (free with host_free): DMA'd directly by check_bulk_ids / submit."""
        p = C.c_void_p()
        self._check(self._L.acl_host_alloc(self._h, int(nbytes), C.byref(p)))
        buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
        a = np.frombuffer(buf, dtype=np.uint8)
        a.flags.writeable = True
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if p is not None and self._h:
            self._check(self._L.acl_host_free(self._h, p))

    def submit_ids(self, items: np.ndarray, perm: np.ndarray, err: np.ndarray):
        """acl_check_bulk_ids_submit: returns a ticket; the arrays must stay alive (and untouched) until wait()."""
        t = C.c_void_p()
        self._check(self._L.acl_check_bulk_ids_submit(self._h, items.ctypes.data, items.size, perm.ctypes.data, err.ctypes.data, C.byref(t)))
        return t

    def wait(self, ticket):
        self._check(self._L.acl_ticket_wait(self._h, ticket))

    def check_bulk_ids_device(self, d_items: int, n: int, d_perm: int, d_err: int = 0):
        """Device pointers (ints); asynchronous after return only w.r.t. the output buffers' readers on acl_stream."""
        self._check(self._L.acl_check_bulk_ids_device(self._h, d_items, n, d_perm, d_err or None))

    def sync(self):
        self._check(self._L.acl_sync(self._h))

    @property
    def stream(self) -> int:
        return self._L.acl_stream(self._h) or 0

    # ---- callers either side of the kernels
    def check_bulk_keep(self, items, item_off):
        """filterItemsWithBulkPermissions (postfilter.go:58-182): items = all resolved pairs [(rt, rid, perm, st, sid, srel)],
        pairs [item_off[i], item_off[i+1]) belong to list item i -> keep mask (uint8[K])."""
        n = len(items)
        arr = (CheckItem * max(1, n))()
        for i, it in enumerate(items):
            arr[i] = CheckItem(*[_b(x if x is not None else "") for x in it])
        off = np.ascontiguousarray(item_off, dtype=np.uint32)
        keep = np.zeros(max(1, off.size - 1), dtype=np.uint8)
        self._check(self._L.acl_check_bulk_keep(self._h, arr, n, off.ctypes.data, off.size - 1, keep.ctypes.data))
        return keep[:off.size - 1]

    def check_bulk_keep_views(self, prepared, item_off):
        """acl_check_bulk_keep_v: the pairs as {pointer, length} views (make_check_views); a call whose pairs share type, permission and subject is
        answered by one reverse walk + bit tests (stats()["keep_route_calls"])."""
        views, n, _blob = prepared
        off = np.ascontiguousarray(item_off, dtype=np.uint32)
        keep = np.zeros(max(1, off.size - 1), dtype=np.uint8)
        self._check(self._L.acl_check_bulk_keep_v(self._h, views.ctypes.data, n, off.ctypes.data, off.size - 1, keep.ctypes.data))
        return keep[:off.size - 1]

    @staticmethod
    def make_check_packed(items):
        """An acl_packed_request_t of [(rt, rid, perm, st, sid, srel)]: every distinct string once in a dictionary, six u32 indices per item (an absent
        subject relation: ACL_PACKED_NONE) -- what the cgo shim fills while it walks a kube list.  -> (request struct, the arrays it points into)"""
        n = len(items)
        where, blob, offsets = {}, bytearray(), [0]
        idx = np.full((max(1, n), 6), 0xFFFFFFFF, dtype=np.uint32)
        for i, it in enumerate(items):
            for f, x in enumerate(it):
                b = _b(x if x is not None else "") or b""
                if f == 5 and not b:
                    continue
                k = where.get(b)
                if k is None:
                    k = where[b] = len(offsets) - 1
                    blob += b
                    offsets.append(len(blob))
                idx[i, f] = k
        buf = np.frombuffer(bytes(blob) or b"\0", dtype=np.uint8).copy()
        offs = np.asarray(offsets, dtype=np.uint32)
        rq = _lib.PackedRequest(buf.ctypes.data, offs.ctypes.data, len(offsets) - 1, 0, idx.ctypes.data, n)
        return rq, (buf, offs, idx)

    def check_bulk_packed(self, prepared):
        rq, _keep_alive = prepared
        n = rq.n_items
        perm = np.zeros(max(1, n), dtype=np.uint8)
        err = np.zeros(max(1, n), dtype=np.int32)
        self._check(self._L.acl_check_bulk_packed(self._h, C.byref(rq), perm.ctypes.data, err.ctypes.data, None))
        return perm[:n], err[:n]

    def check_bulk_keep_packed(self, prepared, item_off):
        rq, _keep_alive = prepared
        off = np.ascontiguousarray(item_off, dtype=np.uint32)
        keep = np.zeros(max(1, off.size - 1), dtype=np.uint8)
        self._check(self._L.acl_check_bulk_keep_packed(self._h, C.byref(rq), off.ctypes.data, off.size - 1, keep.ctypes.data))
        return keep[:off.size - 1]

    def check_bulk_keep_ids(self, items: np.ndarray, item_off):
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        off = np.ascontiguousarray(item_off, dtype=np.uint32)
        keep = np.zeros(max(1, off.size - 1), dtype=np.uint8)
        self._check(self._L.acl_check_bulk_keep_ids(self._h, items.ctypes.data, items.size, off.ctypes.data, off.size - 1, keep.ctypes.data))
        return keep[:off.size - 1]

    def check_bulk_keep_ids_device(self, d_items: int, n: int, d_item_off: int, k_items: int, d_keep: int):
        self._check(self._L.acl_check_bulk_keep_ids_device(self._h, d_items, n, d_item_off, k_items, d_keep))

    def filter_list_response(self, body: bytes, templates, user_name: str, request=None):
        """filterListResponse (postfilter.go:17-55) on the list response's bytes -> (filtered body, kept, total).
        request = (name, namespace, resource) of the kube request (rules.NewResolveInput's fallbacks, rules.go:315-342), or None."""
        from ._lib import ListRequest
        arr = (C.c_char_p * max(1, len(templates)))(*[_b(t) for t in templates])
        out, n, kept, total = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
        if request is not None:
            rq = ListRequest(_b(request[0] or ""), _b(request[1] or ""), _b(request[2] or ""))
            self._check(self._L.acl_filter_list_response_req(self._h, body, len(body), arr, len(templates), _b(user_name), C.byref(rq), C.byref(out), C.byref(n),
                                                             C.byref(kept), C.byref(total)))
        else:
            self._check(self._L.acl_filter_list_response(self._h, body, len(body), arr, len(templates), _b(user_name), C.byref(out), C.byref(n),
                                                         C.byref(kept), C.byref(total)))
        try:
            return C.string_at(out, n.value), kept.value, total.value
        finally:
            self._L.acl_free(out)

    BODY_LIST, BODY_TABLE, BODY_OBJECT = 0, 1, 2

    def prefilter_response(self, rtype: str, bitmap: np.ndarray, id_template: str, kind: int, body: bytes):
        """filterList / filterTable / filterObject (responsefilterer.go:349-416) on the kube response's bytes, with the LookupResources bitmap
        as the allowed set -> (filtered body, kept, total).  A single object outside the set raises AclError(code 7, "unauthorized")."""
        bm = np.ascontiguousarray(bitmap, dtype=np.uint32)
        out, n, kept, total = C.c_void_p(), C.c_size_t(), C.c_uint64(), C.c_uint64()
        self._check(self._L.acl_prefilter_response(self._h, self.type_id(rtype), bm.ctypes.data, bm.size, _b(id_template), kind, body, len(body), C.byref(out), C.byref(n),
                                                   C.byref(kept), C.byref(total)))
        try:
            return C.string_at(out, n.value), kept.value, total.value
        finally:
            self._L.acl_free(out)

    def bitmap_names(self, rtype: str, bitmap: np.ndarray, block: int = 512, buf_bytes: int = 1 << 16):
        """acl_bitmap_names: the names of a LookupResources bitmap's objects in id order, fetched a block per call (what the shim's stream does)."""
        bm = np.ascontiguousarray(bitmap, dtype=np.uint32)
        buf = C.create_string_buffer(buf_bytes)
        ends = np.zeros(block, dtype=np.uint32)
        cur, n = C.c_uint64(0), C.c_size_t()
        out = []
        while True:
            self._check(self._L.acl_bitmap_names(self._h, self.type_id(rtype), bm.ctypes.data, bm.size, C.byref(cur), buf, buf_bytes, ends.ctypes.data, block, C.byref(n)))
            if not n.value:
                return out
            raw, prev = buf.raw, 0
            for k in range(n.value):
                out.append(raw[prev:int(ends[k])].decode())
                prev = int(ends[k])

    def bitmap_test_names(self, rtype: str, bitmap: np.ndarray, object_ids):
        """prefilterResult.IsAllowed (lookups.go:25-36) over a LookupResources bitmap -> bool array."""
        bm = np.ascontiguousarray(bitmap, dtype=np.uint32)
        n = len(object_ids)
        arr = (C.c_char_p * max(1, n))(*[_b(x) for x in object_ids])
        out = np.zeros(max(1, n), dtype=np.uint8)
        self._check(self._L.acl_bitmap_test_names(self._h, self.type_id(rtype), bm.ctypes.data, bm.size, arr, n, out.ctypes.data))
        return out[:n].astype(bool)

    def watch_poll(self, after_revision: int, types=()):
        """-> (updates [(revision, op, (rt, rid, rel, st, sid, srel))], next cursor).  after_revision=WATCH_FROM_NOW: just the cursor."""
        out = []

        def cb(_u, rev, op, rp):
            r = rp.contents
            out.append((rev, op, (r.resource_type.decode(), r.resource_id.decode(), r.relation.decode(), r.subject_type.decode(),
                                  r.subject_id.decode(), (r.subject_relation or b"").decode())))

        tids = [self.type_id(t) for t in types]
        if any(t < 0 for t in tids):
            raise AclError(ERR_FAILED_PRECONDITION, "unknown object type in watch request")
        arr = (C.c_int * max(1, len(tids)))(*tids)
        cur = C.c_uint64()
        self._check(self._L.acl_watch_poll(self._h, after_revision, arr, len(tids), WATCH_CB(cb), None, C.byref(cur)))
        return out, cur.value

    def watch_wait(self, after_revision: int, types=(), timeout_s=None, cancel=None) -> int:
        """Blocks until the feed holds an update behind the cursor for one of `types` (acl_watch_wait: the blocking half of Watch.Recv,
        watch.go:38) -> the store's revision; raises AclError DEADLINE_EXCEEDED / CANCELLED by timeout_s / cancel (a ctypes c_int32)."""
        tids = [self.type_id(t) for t in types]
        if any(t < 0 for t in tids):
            raise AclError(ERR_FAILED_PRECONDITION, "unknown object type in watch request")
        arr = (C.c_int * max(1, len(tids)))(*tids)
        cur = C.c_uint64()
        opts = CallOpts(C.pointer(cancel) if cancel is not None else None, int((timeout_s or 0) * 1e9))
        self._check(self._L.acl_watch_wait(self._h, after_revision, arr, len(tids), C.byref(opts), C.byref(cur)))
        return cur.value

    def watch_recheck(self, after_revision: int, rtype, perm, stype, sid, srel=""):
        """RunWatch's loop body for a whole poll (watch.go:38-108): -> ([(revision, op, relationship 6-tuple, permissionship, err)], cursor):
        every update of `rtype` behind the cursor with the decision of ONE bulk Check `rtype:<its resource id>#perm@stype:sid[#srel]`."""
        out = []

        def cb(_u, rev, op, rp, perm_, err_):
            r = rp.contents
            out.append((rev, op, (r.resource_type.decode(), r.resource_id.decode(), r.relation.decode(), r.subject_type.decode(), r.subject_id.decode(),
                                  (r.subject_relation or b"").decode()), int(perm_), int(err_)))

        templ = CheckItem(_b(rtype), _b("-"), _b(perm), _b(stype), _b(sid), _b(srel))
        cur = C.c_uint64()
        self._check(self._L.acl_watch_recheck(self._h, after_revision, C.byref(templ), WATCH_CHECK_CB(cb), None, C.byref(cur)))
        return out, cur.value

    def batcher_start(self, max_items: int = 4096, max_wait_us: int = 200):
        self._check(self._L.acl_batcher_start(self._h, max_items, max_wait_us))

    def batcher_stop(self):
        self._check(self._L.acl_batcher_stop(self._h))

    def batcher_stats(self):
        b, i = C.c_uint64(), C.c_uint64()
        self._check(self._L.acl_batcher_stats(self._h, C.byref(b), C.byref(i)))
        return {"batches": b.value, "items": i.value}

    def check_one(self, rt, rid, perm, st, sid, srel="", cancel=None, timeout_s=None):
        """One CheckPermission (watch.go:50); rides the micro-batcher when it is running.  Blocks; releases the GIL."""
        it = CheckItem(*[_b(x if x is not None else "") for x in (rt, rid, perm, st, sid, srel)])
        p, e = C.c_uint8(), C.c_int32()
        o = self._opts(cancel, timeout_s)
        self._check(self._L.acl_check_one_opts(self._h, C.byref(it), C.byref(p), C.byref(e), C.byref(o) if o else None))
        return p.value, e.value

    def check_one_submit(self, rt, rid, perm, st, sid, srel="", tag=0):
        """acl_check_one without a blocked thread: returns at once; the answer arrives, tagged, through check_completions().
        This is the form a cgo shim binds (a goroutine parks on a channel, one poller drains the queue).  Needs a running batcher."""
        it = CheckItem(*[_b(x if x is not None else "") for x in (rt, rid, perm, st, sid, srel)])
        self._check(self._L.acl_check_one_submit(self._h, C.byref(it), int(tag)))

    def check_completions(self, max_items=256, timeout_s=-1.0):
        """Up to max_items answered submissions as (tag, rc, err, perm); blocks while there are none (timeout_s < 0: until one
        arrives, 0: never).  Releases the GIL."""
        buf = (Completion * max(1, max_items))()
        n = C.c_size_t()
        t = -1 if timeout_s < 0 else int(timeout_s * 1e9)
        self._check(self._L.acl_check_completions(self._h, buf, max_items, t, C.byref(n)))
        return [(int(buf[i].tag), int(buf[i].rc), int(buf[i].err), int(buf[i].perm)) for i in range(n.value)]

    def lookup_one_submit(self, rt, perm, st, sid, srel="", tag=0):
        """One LookupResources request WITHOUT a blocked thread (acl_lookup_one_submit): its answer arrives through lookup_completions."""
        self._check(self._L.acl_lookup_one_submit(self._h, _b(rt), _b(perm), _b(st), _b(sid), _b(srel or ""), int(tag)))

    def lookup_completions(self, rtype=None, max_items=64, timeout_s=-1.0):
        """Up to max_items answered lookups as (tag, rc, count, ids) -- ids = set of resource names when `rtype` is given, else the raw u32 bitmap;
        blocks while there are none (timeout_s < 0: until one arrives, 0: never).  Releases the GIL; frees the engine's rows."""
        from ._lib import LookupCompletion
        buf = (LookupCompletion * max(1, max_items))()
        n = C.c_size_t()
        t = -1 if timeout_s < 0 else int(timeout_s * 1e9)
        self._check(self._L.acl_lookup_completions(self._h, buf, max_items, t, C.byref(n)))
        out = []
        for i in range(n.value):
            c = buf[i]
            row = None
            if c.bitmap:
                row = np.ctypeslib.as_array(c.bitmap, shape=(max(1, c.words),)).copy()[:c.words]
                self._L.acl_free(c.bitmap)
            if row is not None and rtype is not None:
                ids = np.flatnonzero(np.unpackbits(row.view(np.uint8), bitorder="little"))
                row = {self.object_name(rtype, int(k)) for k in ids}
            out.append((int(c.tag), int(c.rc), int(c.count), row))
        return out

    def lookup_one(self, rt, perm, st, sid, srel="", cancel=None, timeout_s=None):
        """One LookupResources request (lookups.go:65) -> set of resource ids; concurrent callers with the same (type,
        permission, subject class) share one batched reverse walk while the batcher runs.  Blocks; releases the GIL.
        The result bitmap is engine-owned (acl_lookup_resources_alloc): no caller-side sizing race with concurrent writes."""
        bm, words, cnt = C.POINTER(C.c_uint32)(), C.c_size_t(), C.c_uint64()
        o = self._opts(cancel, timeout_s)
        self._check(self._L.acl_lookup_resources_alloc(self._h, _b(rt), _b(perm), _b(st), _b(sid), _b(srel or ""), C.byref(o) if o else None,
                                                       C.byref(bm), C.byref(words), C.byref(cnt)))
        try:
            a = np.ctypeslib.as_array(bm, shape=(words.value,)).copy()
        finally:
            self._L.acl_free(bm)
        ids = np.flatnonzero(np.unpackbits(a.view(np.uint8), bitorder="little"))
        return {self.object_name(rt, int(i)) for i in ids}

    def batcher_lookup_stats(self):
        w, n = C.c_uint64(), C.c_uint64()
        self._check(self._L.acl_batcher_lookup_stats(self._h, C.byref(w), C.byref(n)))
        return {"walks": w.value, "lookups": n.value}

    # ---- lookups
    def lookup_bitmap(self, rt, perm, st, sid, srel=""):
        words = (self.object_count(rt) + 1 + 31) // 32 + 1  # +1: the subject may be interned by the call
        bm = np.zeros(words, dtype=np.uint32)
        cnt = C.c_uint64()
        self._check(self._L.acl_lookup_resources(self._h, _b(rt), _b(perm), _b(st), _b(sid), _b(srel or ""), bm.ctypes.data, words, C.byref(cnt)))
        return bm, cnt.value

    def lookup(self, rt, perm, st, sid, srel=""):
        """-> set of resource ids (strings), as pkg/authz/lookups.go:129 collects them."""
        bm, _ = self.lookup_bitmap(rt, perm, st, sid, srel)
        ids = np.flatnonzero(np.unpackbits(bm.view(np.uint8), bitorder="little"))
        return {self.object_name(rt, int(i)) for i in ids}

    def lookup_ids_batch(self, rtype, perm, stype, srel, subject_ids, out=None):
        """n subjects of one class -> (bitmaps [n, words] u32, counts [n] u64).  `out` = a (bitmaps, counts) pair from an earlier call
        to write into (a caller in a loop: a fresh 0.8 MB array per call is ~200 page faults); the engine writes every word."""
        sids = np.ascontiguousarray(subject_ids, dtype=np.uint32)
        words = (self.object_count(rtype) + 31) // 32
        if out is not None and out[0].shape == (sids.size, max(1, words)) and out[0].dtype == np.uint32 and out[0].flags.c_contiguous \
                and out[1].shape == (sids.size,) and out[1].dtype == np.uint64:
            bms, counts = out
        else:
            bms = np.zeros((sids.size, max(1, words)), dtype=np.uint32)
            counts = np.zeros(sids.size, dtype=np.uint64)
        self._check(self._L.acl_lookup_resources_batch(self._h, self.type_id(rtype), self.relation_id(rtype, perm), self.type_id(stype),
                                                       self.relation_id(stype, srel), sids.ctypes.data, sids.size, bms.ctypes.data, max(1, words),
                                                       counts.ctypes.data))
        return bms, counts

    def lookup_ids(self, rtype, perm, stype, srel, subject_id):
        bms, _ = self.lookup_ids_batch(rtype, perm, stype, srel, [subject_id])
        return np.flatnonzero(np.unpackbits(bms[0].view(np.uint8), bitorder="little")).astype(np.uint32)

    def selfcheck_snapshot(self) -> bool:
        """Test hook (store-only engines): update + verify the host snapshot; True when the update was an in-place patch."""
        return self.selfcheck_snapshot_code() == 1

    def selfcheck_snapshot_code(self) -> int:
        """... 1: patched in place, 0: rebuilt, 2: the snapshot was current already."""
        p = C.c_int()
        self._check(self._L.acl_selfcheck_snapshot(self._h, C.byref(p)))
        return int(p.value)

    def selfcheck_json_array(self, body: bytes, arr_open: int, chunk_bytes: int = 0):
        """Test hook: the element spans of the JSON array at body[arr_open] as the list filters find them -> ([(begin, end)], offset of `]`)."""
        cap = max(1, len(body))
        spans = np.zeros(2 * cap, dtype=np.uint64)
        n, close = C.c_size_t(), C.c_size_t()
        self._check(self._L.acl_selfcheck_json_array(self._h, body, len(body), arr_open, chunk_bytes, spans.ctypes.data, cap, C.byref(n), C.byref(close)))
        return [(int(spans[2 * i]), int(spans[2 * i + 1])) for i in range(n.value)], close.value

    def selfcheck_compaction(self, phase: int) -> bool:
        """Test hook (store-only engines): phase 0 = build from a copy-on-write view; phase 1 = catch up, adopt, verify.
        Returns True when phase 1 adopted the background build."""
        a = C.c_int(1)
        self._check(self._L.acl_selfcheck_compaction(self._h, int(phase), C.byref(a)))
        return bool(a.value)

    def replica_calls(self):
        """[(HIP device ordinal, evaluations handed to that replica since open)]"""
        calls, devs = (C.c_uint64 * 64)(), (C.c_int32 * 64)()
        n = self._L.acl_replica_calls(self._h, calls, devs, 64)
        return [(int(devs[i]), int(calls[i])) for i in range(n)]

    # ---- measurement
    def stats(self) -> dict:
        s = Stats()
        self._check(self._L.acl_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in Stats._fields_}

    def stats_reset(self):
        self._check(self._L.acl_stats_reset(self._h))

    def set_timing(self, on: bool):
        self._check(self._L.acl_set_timing(self._h, 1 if on else 0))
