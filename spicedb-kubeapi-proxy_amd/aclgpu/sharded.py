"""Sharded graph: the north star's multi-GPU configuration.

One process per GPU; every rank holds the rows of the object types with
fnv1a(type) mod world == rank (acl_shard_configure).  A batch of the seam's requests
(CheckBulkPermissions pkg/authz/check.go:48, LookupResources pkg/authz/lookups.go:65)
advances one dispatch level at a time on all ranks:

    step (HIP frontier expansion on the local rows; sub-checks whose rows live on
    another shard are appended to an export buffer)
      -> all-gather of the per-rank counts            (tiny)
      -> all-gather of the export buffers over RCCL   (padded to the largest count)
      -> import: every rank keeps the gathered entries it owns
    ... until no rank produced or exported anything; then per-request results are
    MAX-reduced (Check) or broadcast from the resource type's owner (LookupResources).

This module is the host-side protocol only.  It is written SPMD against two small
interfaces so that the same code runs
  * on real ranks        : GpuShard (C ABI, acl_shard_*)  + TorchComm (torch.distributed: nccl = RCCL on GPUs)
  * as G logical shards on one GPU (tests, `bench.py --sharded`): GpuShard + ThreadComm
  * in the CPU protocol tests (gloo, world_size 2): a test double of the shard + TorchComm.
There is no CPU evaluation path here: GpuShard needs the HIP engine.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import torch

from . import _lib
from .engine import AclError, ERR_INTERNAL, ERR_RESOURCE_EXHAUSTED, Engine

MAX_LEVELS = 50  # dispatch max depth, reference pkg/spicedb/spicedb.go:34
VISIT, EXPAND = 1, 2
ENTRY_WORDS = 4  # one frontier entry = 16 B


# ----------------------------------------------------------------------------- communicators
class TorchComm:
    """torch.distributed process group (backend nccl == RCCL over xGMI on GPUs; gloo on CPU)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device) if device is not None else torch.device("cpu")

    def all_gather_small(self, vals):
        t = torch.tensor(list(vals), dtype=torch.int64, device=self.device)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().reshape(self.world, -1)

    def all_gather(self, out, inp):  # flat views: gloo's all-gather only takes 1-D tensors
        self.dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group)

    def all_to_all(self, recv, send):
        """recv[r] <- rank r's send[self.rank]; tensors of any (matching) sizes, empty ones skipped.  Point-to-point pairs
        (batch_isend_irecv: grouped ncclSend/ncclRecv on RCCL; also available on gloo)."""
        ops = []
        for r in range(self.world):
            if r == self.rank:
                if send[r].numel():
                    recv[r].copy_(send[r])
                continue
            peer = self.dist.get_global_rank(self.group, r) if self.group is not None else r
            if send[r].numel():
                ops.append(self.dist.P2POp(self.dist.isend, send[r].view(-1), peer, self.group))
            if recv[r].numel():
                ops.append(self.dist.P2POp(self.dist.irecv, recv[r].view(-1), peer, self.group))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def all_reduce_max(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

    def broadcast(self, t, src):
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)

    def barrier(self):
        self.dist.barrier(group=self.group)


class ThreadComm:
    """G logical shards inside one process (one thread each) -- same collectives through shared memory.
    Used to run the sharded protocol on a single GPU; tensors may live on the device."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    @staticmethod
    def create(world):
        sh = ThreadComm._Shared(world)
        return [ThreadComm(sh, r) for r in range(world)]

    def __init__(self, shared, rank):
        self._s = shared
        self.rank = rank
        self.world = shared.world

    def _exchange(self, value):
        self._s.slots[self.rank] = value
        self._s.barrier.wait()
        vals = list(self._s.slots)
        self._s.barrier.wait()
        return vals

    def all_gather_small(self, vals):
        return np.asarray(self._exchange(list(vals)), dtype=np.int64)

    def all_gather(self, out, inp):
        if inp.is_cuda:
            torch.cuda.current_stream().synchronize()  # peers read this buffer from their own streams
        parts = self._exchange(inp)
        n = inp.shape[0]
        for r, p in enumerate(parts):
            out[r * n:(r + 1) * n].copy_(p)
        if inp.is_cuda:
            torch.cuda.current_stream().synchronize()
        self._s.barrier.wait()  # nobody overwrites its buffer before every peer has copied it

    def all_to_all(self, recv, send):
        if any(x.is_cuda for x in send):
            torch.cuda.current_stream().synchronize()
        parts = self._exchange(send)  # parts[r] = rank r's send list
        for r in range(self.world):
            if recv[r].numel():
                recv[r].copy_(parts[r][self.rank])
        if any(x.is_cuda for x in recv):
            torch.cuda.current_stream().synchronize()
        self._s.barrier.wait()

    def all_reduce_max(self, t):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        parts = self._exchange(t)
        red = parts[0].clone()
        for p in parts[1:]:
            torch.maximum(red, p, out=red)
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self._s.barrier.wait()
        t.copy_(red)

    def broadcast(self, t, src):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        parts = self._exchange(t)
        val = parts[src].clone() if self.rank != src else None
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self._s.barrier.wait()
        if val is not None:
            t.copy_(val)

    def barrier(self):
        self._s.barrier.wait()


# ----------------------------------------------------------------------------- communicators of the NATIVE loop
# acl_shard_check_bulk (csrc/engine_shard_native.cpp) runs the whole level loop inside libaclgpu.so and calls back only
# for the collective itself.  Production: RCCL inside the library (acl_shard_rccl_*; `RcclNative` hands the ncclUniqueId
# around through torch.distributed).  Single-GPU tests: `ThreadNative`, device-to-device copies between the logical
# shards of one process -- the loop, the kernels and every decision are the same code as with RCCL.
_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise ImportError("libamdhip64.so not loadable")
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return _hip


class ThreadNative:
    """acl_shard_comm_t over a ThreadComm: the all-gather is `world` device-to-device copies on the caller's stream."""

    def __init__(self, comm: "ThreadComm", with_all_to_all: bool = True):
        self.comm = comm
        hip = _hiprt()

        def all_gather(_user, d_send, d_recv, nbytes, stream):
            try:
                hip.hipStreamSynchronize(stream)  # my block is complete before any peer copies it
                sends = comm._exchange(int(d_send))
                for r, src in enumerate(sends):
                    if hip.hipMemcpyAsync(d_recv + r * nbytes, src, nbytes, 3, stream):
                        return 13
                hip.hipStreamSynchronize(stream)
                comm._s.barrier.wait()  # nobody overwrites its send block before every peer has copied it
                return 0
            except Exception:  # noqa: BLE001  (a broken barrier: another logical shard failed)
                return 13

        def all_reduce_max(_user, d_buf, n, stream):
            try:
                hip.hipStreamSynchronize(stream)
                mine = np.empty(n, dtype=np.uint8)
                hip.hipMemcpy(mine.ctypes.data, d_buf, n, 2)
                parts = comm._exchange(mine)
                red = np.maximum.reduce(parts)
                hip.hipMemcpy(d_buf, red.ctypes.data, n, 1)
                comm._s.barrier.wait()
                return 0
            except Exception:  # noqa: BLE001
                return 13

        def all_to_all(_user, d_send, d_recv, nbytes, stream):  # block r of my send -> rank r's recv block `me`
            try:
                hip.hipStreamSynchronize(stream)
                sends = comm._exchange(int(d_send))
                for r, src in enumerate(sends):
                    if hip.hipMemcpyAsync(d_recv + r * nbytes, src + comm.rank * nbytes, nbytes, 3, stream):
                        return 13
                hip.hipStreamSynchronize(stream)
                comm._s.barrier.wait()
                return 0
            except Exception:  # noqa: BLE001
                return 13

        self._cbs = (_lib.ALL_GATHER_CB(all_gather), _lib.ALL_REDUCE_CB(all_reduce_max), _lib.ALL_TO_ALL_CB(all_to_all))  # keep the trampolines alive
        self.struct = _lib.ShardComm(None, self._cbs[0], self._cbs[1], self._cbs[2] if with_all_to_all else _lib.ALL_TO_ALL_CB())


class RcclNative:
    """The library's own RCCL communicator: rank 0 makes the ncclUniqueId, torch.distributed (any backend) hands it around."""

    def __init__(self, shard: "GpuShard", comm: "TorchComm"):
        L, h = shard._L, shard._h
        idb = np.zeros(128, dtype=np.uint8)
        if comm.rank == 0:
            shard.e._check(L.acl_shard_rccl_unique_id(idb.ctypes.data))
        t = torch.from_numpy(idb)
        if comm.device is not None:
            t = t.to(comm.device)
        comm.broadcast(t, 0)
        idb = t.cpu().numpy().copy()
        shard.e._check(L.acl_shard_rccl_init(h, idb.ctypes.data, comm.rank, comm.world))
        self.struct = None  # acl_shard_check_bulk_rccl uses the engine's communicator


class IpcNative:
    """acl_shard_comm_t between PROCESSES sharing one device (tools/ipc_comm.hip -> tools/bin/libaclipc.so): windows of device memory exported
    with hipIpcGetMemHandle and mapped by every peer, a barrier in POSIX shared memory.  Test infrastructure (RCCL refuses two ranks on one
    device; the test boxes have one): separate address spaces, HIP contexts and streams under the same level loops."""

    _lib = None

    @classmethod
    def library(cls, build: bool = True):
        if cls._lib is None:
            root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            so = os.path.join(root, "tools", "bin", "libaclipc.so")
            src = os.path.join(root, "tools", "ipc_comm.hip")
            if build and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)):
                os.makedirs(os.path.dirname(so), exist_ok=True)
                subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-I", os.path.join(root, "include"),
                                       "-lrt", "-o", so])
            L = C.CDLL(so)
            L.aclipc_last_error.restype = C.c_char_p
            L.aclipc_open.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
            L.aclipc_comm.argtypes = [C.c_void_p, C.POINTER(_lib.ShardComm)]
            L.aclipc_comm.restype = None
            L.aclipc_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
            L.aclipc_stats.restype = None
            L.aclipc_barrier.argtypes = [C.c_void_p]
            L.aclipc_close.argtypes = [C.c_void_p]
            L.aclipc_close.restype = None
            cls._lib = L
        return cls._lib

    def __init__(self, name: str, rank: int, world: int, device: int = 0, window_bytes: int = 8 << 20, deadline_s: int = 60, with_all_to_all: bool = True):
        L = self.library()
        self.rank, self.world = rank, world
        self._c = C.c_void_p()
        if L.aclipc_open(name.encode(), rank, world, device, window_bytes, deadline_s, C.byref(self._c)):
            raise AclError(ERR_INTERNAL, (L.aclipc_last_error() or b"").decode())
        self.struct = _lib.ShardComm()
        L.aclipc_comm(self._c, C.byref(self.struct))
        if not with_all_to_all:
            self.struct.all_to_all = _lib.ALL_TO_ALL_CB()

    def barrier(self):
        if self.library().aclipc_barrier(self._c):
            raise AclError(ERR_INTERNAL, (self.library().aclipc_last_error() or b"").decode())

    def stats(self):
        out = (C.c_uint64 * 4)()
        self.library().aclipc_stats(self._c, out)
        return {"collectives": int(out[0]), "foreign_bytes": int(out[1]), "barriers": int(out[2]), "longest_barrier_wait_ms": out[3] / 1e6}

    def close(self):
        if self._c:
            self.library().aclipc_close(self._c)
            self._c = C.c_void_p()


# ----------------------------------------------------------------------------- one shard on one GPU
class GpuShard:
    """One shard of the graph on one MI355X: an Engine configured with (rank, world), stepped through the
    acl_shard_* entry points of the C ABI (include/aclgpu.h).  Buffers are torch tensors on the engine's device;
    the engine's HIP stream is made torch's current stream so kernels and collectives are ordered."""

    def __init__(self, engine: Engine, rank: int, world: int):
        self.e = engine
        self._L = engine._L
        self._h = engine._h
        self.rank, self.world = rank, world
        engine._check(self._L.acl_shard_configure(self._h, rank, world))
        self.device = torch.device("cuda", torch.cuda.current_device())

    @contextlib.contextmanager
    def stream(self):
        ext = torch.cuda.ExternalStream(self._L.acl_shard_stream(self._h) or 0, device=self.device)
        with torch.cuda.stream(ext):
            yield

    def owner_of_type(self, t: str) -> int:
        return self._L.acl_shard_of_type(self._h, self.e.type_id(t))

    def grow_frontier(self):
        self.e._check(self._L.acl_shard_grow_frontier(self._h))

    def _step(self, fn, *args):
        st = _lib.ShardStep()
        self.e._check(fn(self._h, *args, C.byref(st)))
        return int(st.exported), int(st.produced), int(st.overflow)

    # -- Check
    def check_begin(self, items, has, err):
        self.e._check(self._L.acl_shard_check_begin(self._h, items.data_ptr(), items.numel() * items.element_size() // 16, has.data_ptr(),
                                                    err.data_ptr()))

    def check_step(self, level, has, err, export):
        return self._step(self._L.acl_shard_check_step, level, has.data_ptr(), err.data_ptr(), export.data_ptr(), export.shape[0])

    def check_step_by_dest(self, level, has, err, export, cap_per_dest):
        """all-to-all form: `export` holds `world` buffers of cap_per_dest entries -> ((largest count, produced, overflow), counts[world])"""
        st = _lib.ShardStep()
        counts = (C.c_uint64 * self.world)()
        self.e._check(self._L.acl_shard_check_step_by_dest(self._h, level, has.data_ptr(), err.data_ptr(), export.data_ptr(), cap_per_dest, C.byref(st), counts))
        return (int(st.exported), int(st.produced), int(st.overflow)), [int(c) for c in counts]

    def check_import(self, level, entries, n):
        self.e._check(self._L.acl_shard_check_import(self._h, level, entries.data_ptr(), n))

    def check_finish(self, has, err, perm, errout):
        self.e._check(self._L.acl_shard_check_finish(self._h, has.data_ptr(), err.data_ptr(), has.numel(), perm.data_ptr(), errout.data_ptr()))

    # -- LookupResources
    def lookup_begin(self, rtype, perm, stype, srel, sids):
        sids = np.ascontiguousarray(sids, dtype=np.uint32)
        self.e._check(self._L.acl_shard_lookup_begin(self._h, self.e.type_id(rtype), self.e.relation_id(rtype, perm), self.e.type_id(stype),
                                                     self.e.relation_id(stype, srel), sids.ctypes.data, sids.size))

    def lookup_step(self, it, phase, export):
        return self._step(self._L.acl_shard_lookup_step, it, phase, export.data_ptr(), export.shape[0])

    def lookup_import(self, it, entries, n):
        self.e._check(self._L.acl_shard_lookup_import(self._h, it, entries.data_ptr(), n))

    def lookup_words(self, rtype):
        return max(1, (self.e.object_count(rtype) + 31) // 32)

    def lookup_finish(self, bitmaps):
        self.e._check(self._L.acl_shard_lookup_finish(self._h, bitmaps.data_ptr(), bitmaps.shape[1]))


# ----------------------------------------------------------------------------- the SPMD protocol
class _Redo(Exception):
    """capacity was short on some rank: every rank restarts the batch (decided from all-gathered facts)"""


class ShardedEngine:
    """SPMD driver: call the same method with the same arguments on every rank."""

    def __init__(self, shard, comm, export_entries: int = 1 << 16, exchange: str = "allgather", native=None):
        """exchange: how Check frontiers cross shards -- "allgather" (the north star's form: every rank receives every export
        buffer and keeps what it owns) or "alltoall" (exports grouped by owner on the device, each rank receives only its own:
        G times fewer bytes).  This host-driven protocol's LookupResources always all-gathers (a visited state goes to every
        shard that holds parent rows for it); the native loop (lookup_ids_batch_native) sends it to exactly those shards."""
        assert shard.rank == comm.rank and shard.world == comm.world
        assert exchange in ("allgather", "alltoall")
        self.shard, self.comm, self.exchange = shard, comm, exchange
        self._native = native  # the native loops' communicator (default: made from `comm` on first use)
        self.cap = 0
        self._alloc(export_entries)
        self.levels_last = 0
        self.exchanged_entries = 0  # entries this rank received from all-gathers since construction
        self.exchanges = 0

    def _alloc(self, entries):
        entries = max(8, (int(entries) + 7) // 8 * 8)  # multiples of 8 entries (128 B)
        if getattr(self, "exchange", "allgather") == "alltoall":  # one buffer of >= 8 entries per destination
            entries = max(entries, 8 * self.comm.world)
            entries = (entries + 8 * self.comm.world - 1) // (8 * self.comm.world) * (8 * self.comm.world)
        dev = self.shard.device
        self.export = torch.empty((entries, ENTRY_WORDS), dtype=torch.int32, device=dev)
        self.gather = torch.empty((entries * self.comm.world, ENTRY_WORDS), dtype=torch.int32, device=dev)
        self.cap = entries

    def _settle(self, info):
        """info[r] = (exported, produced, overflow) of every rank after a step -> largest export count."""
        if (info[:, 2] == 2).any():
            raise AclError(ERR_RESOURCE_EXHAUSTED, "a relationship row exceeds the per-task enumeration limit")
        redo = False
        if (info[:, 2] == 1).any():
            if info[self.comm.rank, 2] == 1:
                self.shard.grow_frontier()
            redo = True
        mx = int(info[:, 0].max())
        if mx > self.cap:
            self._alloc(2 * mx)
            redo = True
        if redo:
            raise _Redo()
        return mx

    def _exchange(self, info, mx, importer, it):
        seg = (mx + 7) // 8 * 8
        w = self.comm.world
        self.comm.all_gather(self.gather[:w * seg], self.export[:seg])
        self.exchanges += 1
        for r in range(w):
            cnt = int(info[r, 0])
            if r != self.comm.rank and cnt:
                importer(it, self.gather[r * seg:r * seg + cnt], cnt)
                self.exchanged_entries += cnt

    # ---- CheckBulkPermissions
    def check_bulk_ids(self, items):
        """items: acl_item_t batch (numpy structured array / torch byte tensor), the SAME on every rank.
        -> (perm uint8 tensor, err int32 tensor) on the shard's device, identical on every rank."""
        sh = self.shard
        with sh.stream() if hasattr(sh, "stream") else contextlib.nullcontext():
            if isinstance(items, np.ndarray):
                items = torch.from_numpy(np.ascontiguousarray(items).view(np.uint8).reshape(-1).copy())
            items = items.to(sh.device).contiguous()
            n = items.numel() * items.element_size() // 16
            has = torch.zeros(max(n, 1), dtype=torch.uint8, device=sh.device)
            err = torch.zeros(max(n, 1), dtype=torch.uint8, device=sh.device)
            for _attempt in range(12):
                try:
                    self._check_levels(items, has, err)
                    break
                except _Redo:
                    continue
            else:
                raise AclError(ERR_RESOURCE_EXHAUSTED, "sharded check: capacity still short after 12 retries")
            self.comm.all_reduce_max(has)
            self.comm.all_reduce_max(err)
            perm = torch.zeros(max(n, 1), dtype=torch.uint8, device=sh.device)
            errout = torch.zeros(max(n, 1), dtype=torch.int32, device=sh.device)
            if n:
                sh.check_finish(has[:n], err[:n], perm, errout)
            if perm.is_cuda:
                torch.cuda.current_stream().synchronize()
            return perm[:n], errout[:n]

    def check_bulk_ids_native(self, items):
        """The same answers through acl_shard_check_bulk: the whole level loop inside libaclgpu.so -- one fixed-capacity all-gather
        per level, decisions on the device, one host synchronisation per burst of levels.  -> (perm, err, stats dict)."""
        sh = self.shard
        if getattr(self, "_native", None) is None:
            self._native = RcclNative(sh, self.comm) if isinstance(self.comm, TorchComm) else ThreadNative(self.comm)
        with sh.stream():
            if isinstance(items, np.ndarray):
                items = torch.from_numpy(np.ascontiguousarray(items).view(np.uint8).reshape(-1).copy())
            items = items.to(sh.device).contiguous()
            n = items.numel() * items.element_size() // 16
            perm = torch.zeros(max(n, 1), dtype=torch.uint8, device=sh.device)
            errout = torch.zeros(max(n, 1), dtype=torch.int32, device=sh.device)
            torch.cuda.current_stream().synchronize()
            st = _lib.ShardBulkStats()
            if self._native.struct is None:
                rc = sh._L.acl_shard_check_bulk_rccl(sh._h, items.data_ptr(), n, perm.data_ptr(), errout.data_ptr(), C.byref(st))
            else:
                rc = sh._L.acl_shard_check_bulk(sh._h, C.byref(self._native.struct), items.data_ptr(), n, perm.data_ptr(), errout.data_ptr(), C.byref(st))
            sh.e._check(rc)
            self.levels_last = int(st.levels)
            return perm[:n], errout[:n], {f: int(getattr(st, f)) for f, _t in _lib.ShardBulkStats._fields_}

    def lookup_ids_batch_native(self, rtype, perm, stype, srel, subject_ids):
        """LookupResources through acl_shard_lookup_bulk: the whole reverse level loop inside libaclgpu.so.  -> (int32 tensor [n, words], stats)"""
        sh = self.shard
        if getattr(self, "_native", None) is None:
            self._native = RcclNative(sh, self.comm) if isinstance(self.comm, TorchComm) else ThreadNative(self.comm)
        sids = np.ascontiguousarray(subject_ids, dtype=np.uint32)
        e = sh.e
        with sh.stream():
            words = sh.lookup_words(rtype)
            bitmaps = torch.zeros((max(1, sids.size), words), dtype=torch.int32, device=sh.device)
            torch.cuda.current_stream().synchronize()
            st = _lib.ShardBulkStats()
            args = (e.type_id(rtype), e.relation_id(rtype, perm), e.type_id(stype), e.relation_id(stype, srel), sids.ctypes.data, sids.size, bitmaps.data_ptr(), words,
                    C.byref(st))
            if self._native.struct is None:
                rc = sh._L.acl_shard_lookup_bulk_rccl(sh._h, *args)
            else:
                rc = sh._L.acl_shard_lookup_bulk(sh._h, C.byref(self._native.struct), *args)
            e._check(rc)
            return bitmaps[:sids.size], {f: int(getattr(st, f)) for f, _t in _lib.ShardBulkStats._fields_}

    def _check_levels(self, items, has, err):
        if self.exchange == "alltoall":
            return self._check_levels_a2a(items, has, err)
        sh = self.shard
        sh.check_begin(items, has, err)
        for level in range(1, MAX_LEVELS + 1):
            rep = sh.check_step(level, has, err, self.export)
            info = self.comm.all_gather_small(rep)
            mx = self._settle(info)
            self.levels_last = level
            if mx == 0 and not info[:, 1].any():
                break
            if mx:
                self._exchange(info, mx, sh.check_import, level)

    def _check_levels_a2a(self, items, has, err):
        """Same level loop with exports grouped by destination: export = world buffers of `cap // world` entries."""
        sh, w, me = self.shard, self.comm.world, self.comm.rank
        per = self.cap // w  # _alloc keeps cap a multiple of 8 * world
        sh.check_begin(items, has, err)
        for level in range(1, MAX_LEVELS + 1):
            rep, counts = sh.check_step_by_dest(level, has, err, self.export, per)
            info = self.comm.all_gather_small(list(rep) + counts)  # [r] = (largest, produced, overflow, count to 0, ..., count to w-1)
            if (info[:, 2] == 2).any():
                raise AclError(ERR_RESOURCE_EXHAUSTED, "a relationship row exceeds the per-task enumeration limit")
            redo = False
            if (info[:, 2] == 1).any():
                if info[me, 2] == 1:
                    sh.grow_frontier()
                redo = True
            mx = int(info[:, 3:].max())
            if mx > per:
                self._alloc(2 * mx * w)
                redo = True
            if redo:
                raise _Redo()
            self.levels_last = level
            total = int(info[:, 3:].sum())
            if total == 0 and not info[:, 1].any():
                break
            if total:
                send = [self.export[d * per:d * per + int(info[me, 3 + d])] for d in range(w)]
                off = np.concatenate([[0], np.cumsum(info[:, 3 + me])]).astype(np.int64)
                if int(off[-1]) > self.gather.shape[0]:
                    self.gather = torch.empty((int(off[-1]) * 2, ENTRY_WORDS), dtype=torch.int32, device=sh.device)
                recv = [self.gather[int(off[r]):int(off[r + 1])] for r in range(w)]
                self.comm.all_to_all(recv, send)
                self.exchanges += 1
                n_in = int(off[-1])
                if n_in:
                    sh.check_import(level, self.gather[:n_in], n_in)
                    self.exchanged_entries += n_in - int(info[me, 3 + me])

    # ---- LookupResources
    def lookup_ids_batch(self, rtype, perm, stype, srel, subject_ids):
        """-> int32 tensor [n, words]: bit id of row i set <=> rtype:id # perm @ stype:subject_ids[i]#srel (every rank)."""
        sh = self.shard
        sids = np.ascontiguousarray(subject_ids, dtype=np.uint32)
        with sh.stream() if hasattr(sh, "stream") else contextlib.nullcontext():
            for _attempt in range(12):
                try:
                    self._lookup_levels(rtype, perm, stype, srel, sids)
                    break
                except _Redo:
                    continue
            else:
                raise AclError(ERR_RESOURCE_EXHAUSTED, "sharded lookup: capacity still short after 12 retries")
            bitmaps = torch.zeros((max(1, sids.size), sh.lookup_words(rtype)), dtype=torch.int32, device=sh.device)
            if sids.size:
                sh.lookup_finish(bitmaps[:sids.size])
            self.comm.broadcast(bitmaps, sh.owner_of_type(rtype))
            if bitmaps.is_cuda:
                torch.cuda.current_stream().synchronize()
            return bitmaps[:sids.size]

    def _lookup_levels(self, rtype, perm, stype, srel, sids):
        sh = self.shard
        sh.lookup_begin(rtype, perm, stype, srel, sids)
        it = 1
        sh.lookup_step(it, EXPAND, self.export)  # the seeds
        while it + 2 < 2 * (MAX_LEVELS + 2):
            it += 1
            rep = sh.lookup_step(it, VISIT, self.export)
            info = self.comm.all_gather_small(rep)
            mx = self._settle(info)
            self.levels_last = it // 2
            if mx == 0 and not info[:, 1].any():
                break
            if mx:
                self._exchange(info, mx, sh.lookup_import, it)
            it += 1
            sh.lookup_step(it, EXPAND, self.export)


def run_logical_shards(world: int, make_shard, fn):
    """Runs `fn(ShardedEngine)` on `world` logical shards, one thread each, with a ThreadComm.
    make_shard(rank, world) -> shard object.  Returns the per-rank results; re-raises the first failure."""
    comms = ThreadComm.create(world)
    out, errs = [None] * world, [None] * world
    exchange = getattr(fn, "exchange", "allgather")
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None

    def work(r):
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            sh = make_shard(r, world)
            out[r] = fn(ShardedEngine(sh, comms[r], exchange=exchange))
        except BaseException as ex:  # noqa: BLE001
            errs[r] = ex
            comms[r]._s.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for ex in errs:
        if ex is not None and not isinstance(ex, threading.BrokenBarrierError):
            raise ex
    for ex in errs:
        if ex is not None:
            raise ex
    return out
