"""ctypes binding for the C oracle (oracle/acl_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libacl_oracle.so")

PERM_UNSPEC, PERM_NO, PERM_HAS, PERM_COND = 0, 1, 2, 3
OP_CREATE, OP_TOUCH, OP_DELETE = 1, 2, 3
PRE_MUST_NOT_MATCH, PRE_MUST_MATCH = 1, 2
ERR_INVALID_ARGUMENT, ERR_ALREADY_EXISTS, ERR_FAILED_PRECONDITION, ERR_DEPTH = 3, 6, 9, 100


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "acl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libacl_oracle.so"])
    return _SO


class _Rel(C.Structure):
    _fields_ = [(n, C.c_char_p) for n in ("rtype", "rid", "rel", "stype", "sid", "srel")] + [("expires_at", C.c_int64)]


class _Update(C.Structure):
    _fields_ = [("op", C.c_int), ("rel", _Rel)]


class _Filter(C.Structure):
    _fields_ = [("op", C.c_int)] + [(n, C.c_char_p) for n in ("rtype", "rid", "rel", "stype", "sid", "srel")]


_READ_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64)

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_new.restype = C.c_void_p
        L.orc_new.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_set_now.argtypes = [C.c_void_p, C.c_int64]
        L.orc_revision.restype = C.c_uint64
        L.orc_revision.argtypes = [C.c_void_p]
        L.orc_num_tuples.restype = C.c_size_t
        L.orc_num_tuples.argtypes = [C.c_void_p]
        L.orc_type_id.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_rel_id.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.orc_object_name.restype = C.c_char_p
        L.orc_object_name.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.orc_counters.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
        L.orc_add_edges.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_freeze.argtypes = [C.c_void_p]
        L.orc_write.argtypes = [C.c_void_p, C.POINTER(_Update), C.c_int, C.POINTER(_Filter), C.c_int, C.POINTER(C.c_uint64)]
        L.orc_delete_by_filter.argtypes = [C.c_void_p, C.POINTER(_Filter), C.POINTER(C.c_uint64)]
        L.orc_read.argtypes = [C.c_void_p, C.POINTER(_Filter), _READ_CB, C.c_void_p]
        L.orc_check.argtypes = [C.c_void_p] + [C.c_char_p] * 6 + [C.POINTER(C.c_int)]
        L.orc_check_bulk_ids.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_check_bulk_ids_mt.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_intern.restype = C.c_uint32
        L.orc_intern.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.orc_tuned_supported.argtypes = [C.c_void_p]
        L.orc_tuned_build.argtypes = [C.c_void_p]
        L.orc_tuned_check_bulk_ids_mt.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_lookup_ids.restype = C.c_long
        L.orc_lookup_ids.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.orc_lookup.restype = C.c_long
        L.orc_lookup.argtypes = [C.c_void_p] + [C.c_char_p] * 5 + [C.POINTER(C.c_int)]
        L.orc_lookup_error.argtypes = [C.c_void_p]
        L.orc_set_lenient_lookup.argtypes = [C.c_void_p, C.c_int]
        L.orc_lookup_result.restype = C.POINTER(C.c_uint32)
        L.orc_lookup_result.argtypes = [C.c_void_p]
        L.orc_check_bytes.restype = C.c_uint64
        L.orc_check_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_int)]
        L.orc_check_bytes_bulk_mt.restype = C.c_uint64
        L.orc_check_bytes_bulk_mt.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _b(s):
    return None if s is None else s.encode()


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def parse_rel(text: str):
    """tuple text grammar of pkg/rules/rules.go:1053-1055 -> 6-tuple of str."""
    import re
    m = re.match(r"^(?P<rt>.*?):(?P<rid>.*?)#(?P<rel>.*?)@(?P<st>.*?):(?P<sid>.*?)(#(?P<srel>.*?))?$", text)
    if not m:
        raise ValueError(f"invalid relationship text {text!r}")
    return (m["rt"], m["rid"], m["rel"], m["st"], m["sid"], m["srel"] or "")


class Oracle:
    def __init__(self, schema: str):
        L = lib()
        err = C.create_string_buffer(512)
        self._h = L.orc_new(schema.encode(), err, 512)
        if not self._h:
            raise OracleError(ERR_INVALID_ARGUMENT, err.value.decode())
        self._L = L

    def close(self):
        if self._h:
            self._L.orc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, code):
        return OracleError(code, self._L.orc_last_error(self._h).decode())

    def set_now(self, t: int):
        self._L.orc_set_now(self._h, int(t))

    @property
    def revision(self):
        return self._L.orc_revision(self._h)

    def num_tuples(self):
        return self._L.orc_num_tuples(self._h)

    # ---- writes
    @staticmethod
    def _mkrel(r, expires=0):
        rt, rid, rel, st, sid, srel = r
        return _Rel(_b(rt), _b(rid), _b(rel), _b(st), _b(sid), _b(srel or ""), int(expires))

    @staticmethod
    def _mkfilter(op, rtype, rid=None, rel=None, stype=None, sid=None, srel=None):
        return _Filter(op, _b(rtype), _b(rid), _b(rel), _b(stype), _b(sid), _b(srel))

    def write(self, updates, preconditions=()):
        """updates: [(op, rel6 | 'text', expires?)], preconditions: [(op, dict(filter))]"""
        ups = (_Update * max(1, len(updates)))()
        for i, u in enumerate(updates):
            op, r = u[0], u[1]
            exp = u[2] if len(u) > 2 else 0
            if isinstance(r, str):
                r = parse_rel(r)
            ups[i] = _Update(op, self._mkrel(r, exp))
        pre = (_Filter * max(1, len(preconditions)))()
        for i, (op, f) in enumerate(preconditions):
            pre[i] = self._mkfilter(op, **f)
        rev = C.c_uint64()
        rc = self._L.orc_write(self._h, ups, len(updates), pre, len(preconditions), C.byref(rev))
        if rc:
            raise self._err(rc)
        return rev.value

    def touch(self, *rels):
        return self.write([(OP_TOUCH, r) for r in rels])

    def delete_by_filter(self, **f):
        n = C.c_uint64()
        flt = self._mkfilter(0, **f)
        rc = self._L.orc_delete_by_filter(self._h, C.byref(flt), C.byref(n))
        if rc:
            raise self._err(rc)
        return n.value

    def read(self, **f):
        out = []

        def cb(_u, rt, rid, rel, st, sid, srel, exp):
            out.append((rt.decode(), rid.decode(), rel.decode(), st.decode(), sid.decode(), srel.decode(), exp))

        flt = self._mkfilter(0, **f)
        rc = self._L.orc_read(self._h, C.byref(flt), _READ_CB(cb), None)
        if rc:
            raise self._err(rc)
        return out

    # ---- reads
    def check(self, rtype, rid, perm, stype, sid, srel=""):
        """-> (permissionship, err_code)"""
        err = C.c_int()
        p = self._L.orc_check(self._h, _b(rtype), _b(rid), _b(perm), _b(stype), _b(sid), _b(srel), C.byref(err))
        return p, err.value

    def check_text(self, text: str):
        return self.check(*parse_rel(text))

    def lookup(self, rtype, perm, stype, sid, srel=""):
        err = C.c_int()
        n = self._L.orc_lookup(self._h, _b(rtype), _b(perm), _b(stype), _b(sid), _b(srel), C.byref(err))
        if n < 0:
            raise self._err(err.value)
        ids = self._L.orc_lookup_result(self._h)
        t = self._L.orc_type_id(self._h, _b(rtype))
        return {self._L.orc_object_name(self._h, t, ids[i]).decode() for i in range(n)}

    # ---- numeric bulk mode
    def type_id(self, t):
        return self._L.orc_type_id(self._h, _b(t))

    def rel_id(self, t, r):
        return -1 if not r else self._L.orc_rel_id(self._h, self.type_id(t), _b(r))

    def add_edges(self, rtype, rel, stype, srel, res, subj):
        res = np.ascontiguousarray(res, dtype=np.uint32)
        subj = np.ascontiguousarray(subj, dtype=np.uint32)
        assert res.shape == subj.shape
        # srel "*": `stype:*` relationships (one per entry of res; subj is ignored)
        rc = self._L.orc_add_edges(self._h, self.type_id(rtype), self.rel_id(rtype, rel), self.type_id(stype),
                                   -2 if srel == "*" else self.rel_id(stype, srel), res.size, res.ctypes.data, subj.ctypes.data)
        if rc:
            raise self._err(rc)

    def freeze(self):
        self._L.orc_freeze(self._h)

    def check_bulk_ids(self, rtype, perm, res, stype, srel, subj):
        res = np.ascontiguousarray(res, dtype=np.uint32)
        subj = np.ascontiguousarray(subj, dtype=np.uint32)
        out = np.zeros(res.size, dtype=np.uint8)
        err = np.zeros(res.size, dtype=np.int32)
        self._L.orc_check_bulk_ids(self._h, res.size, self.type_id(rtype), self.rel_id(rtype, perm), res.ctypes.data,
                                   self.type_id(stype), self.rel_id(stype, srel), subj.ctypes.data, out.ctypes.data, err.ctypes.data)
        return out, err

    def check_bulk_ids_mt(self, nthreads, rtype, perm, res, stype, srel, subj):
        """Same answers as check_bulk_ids, the batch split statically over `nthreads` host threads."""
        res = np.ascontiguousarray(res, dtype=np.uint32)
        subj = np.ascontiguousarray(subj, dtype=np.uint32)
        out = np.zeros(res.size, dtype=np.uint8)
        err = np.zeros(res.size, dtype=np.int32)
        self._L.orc_check_bulk_ids_mt(self._h, int(nthreads), res.size, self.type_id(rtype), self.rel_id(rtype, perm), res.ctypes.data,
                                      self.type_id(stype), self.rel_id(stype, srel), subj.ctypes.data, out.ctypes.data, err.ctypes.data)
        return out, err

    def intern(self, t, oid) -> int:
        """the dense id of object `oid` of type `t` (created if missing): what the numeric entry points take"""
        return self._L.orc_intern(self._h, self.type_id(t), _b(oid))

    def tuned_build(self) -> bool:
        """builds the tuned evaluator's row index (setup, not evaluation); False: the schema / data are outside what it takes (`&`, `-`, `.all()`, wildcards)"""
        return self._L.orc_tuned_build(self._h) == 0

    def tuned_check_bulk_ids_mt(self, nthreads, rtype, perm, res, stype, srel, subj):
        """The tuned CPU Check (row index, level-synchronous frontier with merged states, dynamic chunks over `nthreads`): same answers as check_bulk_ids --
        every caller asserts that -- or None when the schema is outside what it takes."""
        res = np.ascontiguousarray(res, dtype=np.uint32)
        subj = np.ascontiguousarray(subj, dtype=np.uint32)
        out = np.zeros(res.size, dtype=np.uint8)
        err = np.zeros(res.size, dtype=np.int32)
        rc = self._L.orc_tuned_check_bulk_ids_mt(self._h, int(nthreads), res.size, self.type_id(rtype), self.rel_id(rtype, perm), res.ctypes.data,
                                                 self.type_id(stype), self.rel_id(stype, srel), subj.ctypes.data, out.ctypes.data, err.ctypes.data)
        return None if rc else (out, err)

    def lookup_ids(self, rtype, perm, stype, srel, subj):
        n = self._L.orc_lookup_ids(self._h, self.type_id(rtype), self.rel_id(rtype, perm), self.type_id(stype),
                                   self.rel_id(stype, srel), int(subj))
        if n < 0:  # a candidate's Check erred: the reference's stream fails (pkg/authz/lookups.go:75-83)
            raise self._err(self._L.orc_lookup_error(self._h))
        ids = self._L.orc_lookup_result(self._h)
        return np.ctypeslib.as_array(ids, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint32)

    def set_lenient_lookup(self, on: bool):
        """on: a candidate whose Check errs is dropped from the answer instead of failing the lookup (the engine's ACL_FLAG_LENIENT_LOOKUP)"""
        self._L.orc_set_lenient_lookup(self._h, int(bool(on)))

    def check_bytes(self, rtype, perm, res, stype, srel, subj):
        r = C.c_int()
        b = self._L.orc_check_bytes(self._h, self.type_id(rtype), self.rel_id(rtype, perm), int(res), self.type_id(stype),
                                    self.rel_id(stype, srel), int(subj), C.byref(r))
        return b, r.value

    def check_bytes_bulk(self, nthreads, rtype, perm, res, stype, srel, subj):
        """SURVEY.md 8(d) algorithmic bytes of a whole batch (multi-threaded) -> (total bytes, bytes per level [51], distinct states per level [51])"""
        res = np.ascontiguousarray(res, dtype=np.uint32)
        subj = np.ascontiguousarray(subj, dtype=np.uint32)
        lb = np.zeros(51, dtype=np.uint64)
        ls = np.zeros(51, dtype=np.uint64)
        tot = self._L.orc_check_bytes_bulk_mt(self._h, int(nthreads), res.size, self.type_id(rtype), self.rel_id(rtype, perm), res.ctypes.data,
                                              self.type_id(stype), self.rel_id(stype, srel), subj.ctypes.data, lb.ctypes.data, ls.ctypes.data)
        return int(tot), lb, ls

    def counters(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._L.orc_counters(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
