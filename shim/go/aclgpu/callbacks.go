package aclgpu

/*
#include "aclgpu.h"
*/
import "C"

import (
	"runtime/cgo"
	"unsafe"

	v1 "github.com/authzed/authzed-go/proto/authzed/api/v1"
)

type readSink struct{ rels []*v1.Relationship }

type watchSink struct{ updates []watchUpdate }

type watchUpdate struct {
	revision uint64
	op       int32
	rel      *v1.Relationship
}

//export goReadCallback
func goReadCallback(user unsafe.Pointer, rel *C.acl_relationship_t) {
	s := cgo.Handle(uintptr(user)).Value().(*readSink)
	s.rels = append(s.rels, relationshipFromC(rel))
}

//export goWatchCallback
func goWatchCallback(user unsafe.Pointer, revision C.uint64_t, op C.int32_t, rel *C.acl_relationship_t) {
	s := cgo.Handle(uintptr(user)).Value().(*watchSink)
	s.updates = append(s.updates, watchUpdate{uint64(revision), int32(op), relationshipFromC(rel)})
}
