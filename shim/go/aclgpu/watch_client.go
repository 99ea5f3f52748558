package aclgpu

/*
#include "shim.h"
*/
import "C"

import (
	"context"
	"io"
	"math"
	"runtime/cgo"
	"time"
	"unsafe"

	v1 "github.com/authzed/authzed-go/proto/authzed/api/v1"
	"google.golang.org/grpc"
	"google.golang.org/grpc/metadata"
	"google.golang.org/grpc/status"
)

type watchClient struct{ e *Engine }

// NewWatchClient is what goes into proxy.Options.WatchClient (pkg/proxy/options.go:81; consumer pkg/authz/watch.go:27-38).
func NewWatchClient(e *Engine) v1.WatchServiceClient { return &watchClient{e} }

// Watch starts at the head revision (the reference passes no start cursor, watch.go:29-31) and polls the engine's
// change feed; one WatchResponse per committed revision.
func (w *watchClient) Watch(ctx context.Context, in *v1.WatchRequest, _ ...grpc.CallOption) (v1.WatchService_WatchClient, error) {
	var cs cstrings
	defer cs.free()
	types := make([]C.int, 0, len(in.OptionalObjectTypes))
	for _, t := range in.OptionalObjectTypes {
		types = append(types, C.acl_type_id(w.e.h, cs.add(t)))
	}
	s := &watchStream{ctx: ctx, e: w.e, types: types}
	var tp *C.int
	if len(types) > 0 {
		tp = &types[0]
	}
	var head C.uint64_t
	if rc := C.acl_watch_poll_go(w.e.h, C.uint64_t(math.MaxUint64), tp, C.int(len(types)), nil, &head); rc != 0 {
		return nil, lastError(rc)
	}
	s.cursor = uint64(head)
	return s, nil
}

type watchStream struct {
	ctx     context.Context
	e       *Engine
	types   []C.int
	cursor  uint64
	pending []*v1.WatchResponse
}

func (s *watchStream) Recv() (*v1.WatchResponse, error) {
	for len(s.pending) == 0 {
		if err := s.ctx.Err(); err != nil {
			return nil, status.FromContextError(err).Err()
		}
		sink := &watchSink{}
		hdl := cgo.NewHandle(sink)
		var tp *C.int
		if len(s.types) > 0 {
			tp = &s.types[0]
		}
		var next C.uint64_t
		rc := C.acl_watch_poll_go(s.e.h, C.uint64_t(s.cursor), tp, C.int(len(s.types)), unsafe.Pointer(uintptr(hdl)), &next)
		hdl.Delete()
		if rc != 0 {
			return nil, lastError(rc)
		}
		s.cursor = uint64(next)
		for _, u := range sink.updates { // group per revision
			n := len(s.pending)
			if n == 0 || s.pending[n-1].ChangesThrough.Token != revToken(u.revision) {
				s.pending = append(s.pending, &v1.WatchResponse{ChangesThrough: &v1.ZedToken{Token: revToken(u.revision)}})
				n++
			}
			op := v1.RelationshipUpdate_OPERATION_TOUCH
			if u.op == C.ACL_OP_DELETE {
				op = v1.RelationshipUpdate_OPERATION_DELETE
			}
			s.pending[n-1].Updates = append(s.pending[n-1].Updates, &v1.RelationshipUpdate{Operation: op, Relationship: u.rel})
		}
		if len(s.pending) == 0 {
			// nothing behind the cursor: block in the engine until the feed moves for one of the watched types (acl_watch_wait -- a
			// condition variable on the write path, as the reference's stream blocks in Recv(), watch.go:38), looking at the context
			// every 200 ms; no sleeping poll per open watch
			opts := C.acl_call_opts_t{timeout_ns: C.int64_t(200 * time.Millisecond)}
			var head C.uint64_t
			if rc := C.acl_watch_wait(s.e.h, C.uint64_t(s.cursor), tp, C.int(len(s.types)), &opts, &head); rc != 0 && rc != C.ACL_ERR_DEADLINE_EXCEEDED {
				return nil, lastError(rc)
			}
		}
	}
	r := s.pending[0]
	s.pending = s.pending[1:]
	return r, nil
}

func revToken(rev uint64) string              { return "aclgpu-" + itoa(rev) }
func (s *watchStream) Header() (metadata.MD, error) { return nil, nil }
func (s *watchStream) Trailer() metadata.MD         { return nil }
func (s *watchStream) CloseSend() error             { return nil }
func (s *watchStream) Context() context.Context     { return s.ctx }
func (s *watchStream) SendMsg(any) error            { return nil }
func (s *watchStream) RecvMsg(any) error            { return io.EOF }

func itoa(v uint64) string {
	if v == 0 {
		return "0"
	}
	var b [20]byte
	i := len(b)
	for v > 0 {
		i--
		b[i] = byte('0' + v%10)
		v /= 10
	}
	return string(b[i:])
}
