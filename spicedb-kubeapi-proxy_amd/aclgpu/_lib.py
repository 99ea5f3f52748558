"""ctypes binding of libaclgpu.so (include/aclgpu.h).

The product path: there is no fallback here.  If the shared library is missing
or no MI355X is usable, loading / opening the engine raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("ACLGPU_LIB") or os.path.join(_PKG, "lib", "libaclgpu.so")  # ACLGPU_LIB: A/B a kernel variant build

# every symbol include/aclgpu.h declares (checked by tests/test_abi.py against the header)
SYMBOLS = [
    "acl_open", "acl_close", "acl_last_error", "acl_load_bootstrap", "acl_type_id", "acl_relation_id", "acl_intern", "acl_find",
    "acl_object_name", "acl_object_count", "acl_write", "acl_delete_by_filter", "acl_read", "acl_add_edges", "acl_revision",
    "acl_set_now", "acl_snapshot", "acl_check_bulk", "acl_check_bulk_v", "acl_check_bulk_ids", "acl_check_bulk_ids_device", "acl_stream", "acl_sync",
    "acl_lookup_resources", "acl_lookup_resources_ids", "acl_lookup_resources_batch", "acl_stats", "acl_stats_reset", "acl_set_timing",
    "acl_shard_configure", "acl_shard_of_type", "acl_shard_grow_frontier", "acl_shard_check_begin", "acl_shard_check_step",
    "acl_shard_check_step_by_dest", "acl_shard_check_import", "acl_shard_check_finish", "acl_shard_lookup_begin", "acl_shard_lookup_step", "acl_shard_lookup_import",
    "acl_shard_lookup_finish",
    "acl_check_bulk_keep", "acl_check_bulk_keep_ids", "acl_check_bulk_keep_ids_device", "acl_bitmap_test_names", "acl_watch_poll",
    "acl_batcher_start", "acl_batcher_stop", "acl_batcher_stats", "acl_check_one", "acl_lookup_one", "acl_batcher_lookup_stats",
    "acl_selfcheck_snapshot",
    "acl_delete_by_filter_pre", "acl_check_bulk_ids_opts", "acl_check_bulk_ids_submit", "acl_ticket_wait", "acl_host_alloc", "acl_host_free",
    "acl_lookup_resources_alloc", "acl_free", "acl_check_one_opts", "acl_lookup_one_opts", "acl_shard_stream", "acl_filter_list_response", "acl_filter_list_response_req", "acl_shard_check_bulk", "acl_shard_rccl_unique_id", "acl_shard_rccl_init",
    "acl_shard_rccl_destroy", "acl_shard_check_bulk_rccl", "acl_shard_lookup_bulk", "acl_shard_lookup_bulk_rccl", "acl_selfcheck_compaction", "acl_check_one_submit", "acl_check_completions",
    "acl_lookup_one_submit", "acl_lookup_completions", "acl_prefilter_response", "acl_open_replicas", "acl_replica_calls", "acl_watch_wait", "acl_watch_recheck", "acl_load_bootstrap_yaml",
    "acl_check_bulk_v_opts", "acl_object_name_copy", "acl_resolve_bulk_v", "acl_check_bulk_packed", "acl_check_bulk_keep_v", "acl_check_bulk_keep_packed", "acl_selfcheck_json_array", "acl_bitmap_names",
]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("frontier_entries", C.c_uint64), ("max_sub_batch", C.c_uint32), ("flags", C.c_uint32),
                ("contexts", C.c_uint32), ("reserved", C.c_uint32)]


class PackedRequest(C.Structure):  # acl_packed_request_t
    _fields_ = [("bytes", C.c_void_p), ("offsets", C.c_void_p), ("n_strings", C.c_uint32), ("reserved", C.c_uint32), ("items", C.c_void_p), ("n_items", C.c_size_t)]


class CallOpts(C.Structure):
    """acl_call_opts_t: `cancel` points at an int32 the caller raises to abandon the call; timeout_ns <= 0 = none."""
    _fields_ = [("cancel", C.POINTER(C.c_int32)), ("timeout_ns", C.c_int64)]


class Relationship(C.Structure):
    _fields_ = [(n, C.c_char_p) for n in ("resource_type", "resource_id", "relation", "subject_type", "subject_id", "subject_relation")] + [
        ("expires_at", C.c_int64)]


class Update(C.Structure):
    _fields_ = [("op", C.c_int32), ("rel", Relationship)]


class Filter(C.Structure):
    _fields_ = [("op", C.c_int32)] + [(n, C.c_char_p) for n in ("resource_type", "resource_id", "relation", "subject_type", "subject_id", "subject_relation")]


class CheckItem(C.Structure):
    _fields_ = [(n, C.c_char_p) for n in ("resource_type", "resource_id", "permission", "subject_type", "subject_id", "subject_relation")]


class Completion(C.Structure):
    """acl_completion_t: one answered acl_check_one_submit."""
    _fields_ = [("tag", C.c_uint64), ("rc", C.c_int32), ("err", C.c_int32), ("perm", C.c_uint8), ("pad", C.c_uint8 * 3)]


class ListRequest(C.Structure):
    """acl_list_request_t: the kube request a list response answers (RequestInfo.Name / .Namespace / .Resource)."""
    _fields_ = [("name", C.c_char_p), ("namespace_", C.c_char_p), ("resource", C.c_char_p)]


class LookupCompletion(C.Structure):
    """acl_lookup_completion_t: one answered acl_lookup_one_submit; `bitmap` is the receiver's (acl_free)."""
    _fields_ = [("tag", C.c_uint64), ("rc", C.c_int32), ("reserved", C.c_uint32), ("count", C.c_uint64), ("words", C.c_size_t), ("bitmap", C.POINTER(C.c_uint32))]


class Stats(C.Structure):
    _fields_ = [("check_items", C.c_uint64), ("check_passes", C.c_uint64), ("expand_launches", C.c_uint64), ("levels_last", C.c_uint64),
                ("frontier_entries", C.c_uint64), ("kernel_ms", C.c_double), ("expand_ms", C.c_double), ("snapshot_edges", C.c_uint64),
                ("snapshot_bytes", C.c_uint64), ("snapshot_builds", C.c_uint64), ("overflow_retries", C.c_uint64), ("snapshot_edges_local", C.c_uint64), ("snapshot_patches", C.c_uint64),
                ("local_ms", C.c_double), ("local_passes", C.c_uint64), ("snapshot_compactions", C.c_uint64),
                ("rev_local_ms", C.c_double), ("rev_local_passes", C.c_uint64), ("lookup_requests", C.c_uint64), ("ids_recycled", C.c_uint64), ("keep_route_calls", C.c_uint64), ("depth_sweeps", C.c_uint64)]


ALL_GATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_REDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


ALL_TO_ALL_CB = ALL_GATHER_CB  # (user, d_send, d_recv, bytes per peer, stream)


class ShardComm(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_CB), ("all_reduce_max_u8", ALL_REDUCE_CB), ("all_to_all", ALL_TO_ALL_CB)]


class ShardBulkStats(C.Structure):
    _fields_ = [("levels", C.c_uint32), ("exchanges", C.c_uint32), ("host_syncs", C.c_uint32), ("retries", C.c_uint32),
                ("exchanged_bytes", C.c_uint64), ("entries_exchanged", C.c_uint64), ("export_capacity", C.c_uint64),
                ("data_exchanges", C.c_uint32), ("reserved", C.c_uint32)]


class ShardStep(C.Structure):
    _fields_ = [("exported", C.c_uint64), ("produced", C.c_uint32), ("overflow", C.c_uint32)]


READ_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Relationship))
WATCH_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(Relationship))
WATCH_CHECK_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(Relationship), C.c_uint8, C.c_int32)

_lib = None


def build(force: bool = False) -> str:
    """hipcc cross-compile for gfx950 (works without a GPU)."""
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(_PKG, "csrc", f)) > os.path.getmtime(LIB_PATH) for f in os.listdir(os.path.join(_PKG, "csrc"))):
        subprocess.check_call(["make", "-C", _PKG, "-s", "-j8", "lib/libaclgpu.so"])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing torch first makes the
    # dynamic linker resolve libaclgpu.so's DT_NEEDED libamdhip64.so.7 to that already-loaded copy, so torch tensors,
    # torch.distributed (RCCL) and the engine share one runtime.  (A Go/C host simply gets /opt/rocm's.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.acl_open.argtypes = [C.POINTER(Config), C.POINTER(H)]
    L.acl_open_replicas.argtypes = [C.POINTER(Config), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(H)]
    L.acl_replica_calls.argtypes = [H, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_uint32]
    L.acl_close.argtypes = [H]
    L.acl_close.restype = None
    L.acl_last_error.restype = C.c_char_p
    L.acl_load_bootstrap.argtypes = [H, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.acl_load_bootstrap_yaml.argtypes = [H, C.c_char_p, C.c_size_t]
    L.acl_type_id.argtypes = [H, C.c_char_p]
    L.acl_relation_id.argtypes = [H, C.c_int, C.c_char_p]
    L.acl_intern.argtypes = [H, C.c_int, C.c_char_p, C.POINTER(C.c_uint32)]
    L.acl_find.argtypes = [H, C.c_int, C.c_char_p, C.POINTER(C.c_uint32)]
    L.acl_object_name.argtypes = [H, C.c_int, C.c_uint32]
    L.acl_object_name.restype = C.c_char_p
    L.acl_object_name_copy.argtypes = [H, C.c_int, C.c_uint32, C.c_char_p, C.c_size_t]
    L.acl_object_name_copy.restype = C.c_int64
    L.acl_bitmap_names.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.acl_object_count.argtypes = [H, C.c_int]
    L.acl_object_count.restype = C.c_uint32
    L.acl_write.argtypes = [H, C.POINTER(Update), C.c_int, C.POINTER(Filter), C.c_int, C.POINTER(C.c_uint64)]
    L.acl_delete_by_filter.argtypes = [H, C.POINTER(Filter), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acl_read.argtypes = [H, C.POINTER(Filter), READ_CB, C.c_void_p]
    L.acl_add_edges.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_revision.argtypes = [H]
    L.acl_revision.restype = C.c_uint64
    L.acl_set_now.argtypes = [H, C.c_int64]
    L.acl_snapshot.argtypes = [H]
    L.acl_check_bulk.argtypes = [H, C.POINTER(CheckItem), C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_check_bulk_v.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_check_bulk_v_opts.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(CallOpts)]
    L.acl_check_bulk_ids.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_resolve_bulk_v.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_check_bulk_ids_device.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_stream.argtypes = [H]
    L.acl_stream.restype = C.c_void_p
    L.acl_sync.argtypes = [H]
    L.acl_lookup_resources.argtypes = [H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.acl_lookup_resources_ids.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.acl_lookup_resources_batch.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.acl_stats.argtypes = [H, C.POINTER(Stats)]
    L.acl_stats_reset.argtypes = [H]
    L.acl_set_timing.argtypes = [H, C.c_int]
    L.acl_check_bulk_keep.argtypes = [H, C.POINTER(CheckItem), C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.acl_check_bulk_keep_ids.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.acl_check_bulk_keep_v.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.acl_check_bulk_packed.argtypes = [H, C.POINTER(PackedRequest), C.c_void_p, C.c_void_p, C.POINTER(CallOpts)]
    L.acl_check_bulk_keep_packed.argtypes = [H, C.POINTER(PackedRequest), C.c_void_p, C.c_size_t, C.c_void_p]
    L.acl_check_bulk_keep_ids_device.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.acl_bitmap_test_names.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.c_void_p]
    L.acl_watch_poll.argtypes = [H, C.c_uint64, C.POINTER(C.c_int), C.c_int, WATCH_CB, C.c_void_p, C.POINTER(C.c_uint64)]
    L.acl_watch_wait.argtypes = [H, C.c_uint64, C.POINTER(C.c_int), C.c_int, C.POINTER(CallOpts), C.POINTER(C.c_uint64)]
    L.acl_watch_recheck.argtypes = [H, C.c_uint64, C.POINTER(CheckItem), WATCH_CHECK_CB, C.c_void_p, C.POINTER(C.c_uint64)]
    L.acl_batcher_start.argtypes = [H, C.c_uint32, C.c_uint32]
    L.acl_batcher_stop.argtypes = [H]
    L.acl_batcher_stats.argtypes = [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acl_check_one.argtypes = [H, C.POINTER(CheckItem), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]
    L.acl_lookup_one.argtypes = [H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.acl_batcher_lookup_stats.argtypes = [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acl_selfcheck_snapshot.argtypes = [H, C.POINTER(C.c_int)]
    L.acl_selfcheck_compaction.argtypes = [H, C.c_int, C.POINTER(C.c_int)]
    L.acl_selfcheck_json_array.argtypes = [H, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.acl_delete_by_filter_pre.argtypes = [H, C.POINTER(Filter), C.POINTER(Filter), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acl_check_bulk_ids_opts.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(CallOpts)]
    L.acl_check_bulk_ids_submit.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.acl_ticket_wait.argtypes = [H, C.c_void_p]
    L.acl_host_alloc.argtypes = [H, C.c_size_t, C.POINTER(C.c_void_p)]
    L.acl_host_free.argtypes = [H, C.c_void_p]
    L.acl_lookup_resources_alloc.argtypes = [H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(CallOpts),
                                             C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
    L.acl_filter_list_response_req.argtypes = [H, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.c_char_p, C.POINTER(ListRequest), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acl_filter_list_response.argtypes = [H, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acl_prefilter_response.argtypes = [H, C.c_int, C.c_void_p, C.c_size_t, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acl_free.argtypes = [C.c_void_p]
    L.acl_free.restype = None
    L.acl_check_one_opts.argtypes = [H, C.POINTER(CheckItem), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(CallOpts)]
    L.acl_check_one_submit.argtypes = [H, C.POINTER(CheckItem), C.c_uint64]
    L.acl_check_completions.argtypes = [H, C.POINTER(Completion), C.c_size_t, C.c_int64, C.POINTER(C.c_size_t)]
    L.acl_lookup_one_submit.argtypes = [H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
    L.acl_lookup_completions.argtypes = [H, C.POINTER(LookupCompletion), C.c_size_t, C.c_int64, C.POINTER(C.c_size_t)]
    L.acl_lookup_one_opts.argtypes = [H, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                                      C.POINTER(CallOpts)]
    L.acl_shard_configure.argtypes = [H, C.c_uint32, C.c_uint32]
    L.acl_shard_of_type.argtypes = [H, C.c_int]
    L.acl_shard_grow_frontier.argtypes = [H]
    L.acl_shard_check_bulk.argtypes = [H, C.POINTER(ShardComm), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(ShardBulkStats)]
    L.acl_shard_rccl_unique_id.argtypes = [C.c_void_p]
    L.acl_shard_rccl_init.argtypes = [H, C.c_void_p, C.c_uint32, C.c_uint32]
    L.acl_shard_rccl_destroy.argtypes = [H]
    L.acl_shard_check_bulk_rccl.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(ShardBulkStats)]
    L.acl_shard_lookup_bulk.argtypes = [H, C.POINTER(ShardComm), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(ShardBulkStats)]
    L.acl_shard_lookup_bulk_rccl.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(ShardBulkStats)]
    L.acl_shard_stream.argtypes = [H]
    L.acl_shard_stream.restype = C.c_void_p
    L.acl_shard_check_begin.argtypes = [H, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_shard_check_step.argtypes = [H, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(ShardStep)]
    L.acl_shard_check_step_by_dest.argtypes = [H, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(ShardStep), C.POINTER(C.c_uint64)]
    L.acl_shard_check_import.argtypes = [H, C.c_uint32, C.c_void_p, C.c_size_t]
    L.acl_shard_check_finish.argtypes = [H, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.acl_shard_lookup_begin.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.acl_shard_lookup_step.argtypes = [H, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(ShardStep)]
    L.acl_shard_lookup_import.argtypes = [H, C.c_uint32, C.c_void_p, C.c_size_t]
    L.acl_shard_lookup_finish.argtypes = [H, C.c_void_p, C.c_size_t]
    _lib = L
    return L
