package aclgpu

/*
#include "shim.h"
*/
import "C"

import (
	"unsafe"
)

// Sharded deployment (the north star's 8-GPU layout, SURVEY.md 8(e)): one process per GPU, each holding the rows of the
// object types with fnv1a(type) mod world == rank; the relationship store stays replicated (every rank applies every
// WriteRelationships).  The level loop runs INSIDE libaclgpu.so (acl_shard_check_bulk_rccl: one fixed-capacity
// ncclAllGather per dispatch level over xGMI, decisions on the device, one host synchronisation per burst of levels).
// UNBUILT here, like the rest of the shim.

// RcclUniqueID is made by rank 0 and handed to every rank by whatever channel the deployment has (the proxy replicas
// already share a control plane); it is the 128-byte ncclUniqueId.
func RcclUniqueID() ([C.ACL_RCCL_UNIQUE_ID_BYTES]byte, error) {
	var id [C.ACL_RCCL_UNIQUE_ID_BYTES]byte
	if rc := C.acl_shard_rccl_unique_id(unsafe.Pointer(&id[0])); rc != 0 {
		return id, lastError(rc)
	}
	return id, nil
}

// JoinShards turns this engine into shard `rank` of `world` and creates the library's RCCL communicator.
func (e *Engine) JoinShards(id [C.ACL_RCCL_UNIQUE_ID_BYTES]byte, rank, world uint32) error {
	if rc := C.acl_shard_rccl_init(e.h, unsafe.Pointer(&id[0]), C.uint32_t(rank), C.uint32_t(world)); rc != 0 {
		return lastError(rc)
	}
	return nil
}

// ShardedCheckBulk answers one batch with all shards together: SPMD, every rank calls it with the SAME interned batch
// (device pointers: the batch is uploaded once per rank by the caller, e.g. through hipMemcpy in a small C helper).
// perm / err come back identical on every rank.
func (e *Engine) ShardedCheckBulk(dItems unsafe.Pointer, n int, dPerm, dErr unsafe.Pointer) (levels, exchanges uint32, err error) {
	var st C.acl_shard_bulk_stats_t
	if rc := C.acl_shard_check_bulk_rccl(e.h, dItems, C.size_t(n), dPerm, dErr, &st); rc != 0 {
		return 0, 0, lastError(rc)
	}
	return uint32(st.levels), uint32(st.exchanges), nil
}
