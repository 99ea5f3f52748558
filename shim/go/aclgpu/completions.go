package aclgpu

/*
#include "shim.h"
*/
import "C"

import (
	"context"
	"runtime"
	"sync"
	"sync/atomic"
	"unsafe"

	"google.golang.org/grpc/codes"
	"google.golang.org/grpc/status"
)

// Single checks without an OS thread blocked in C per request (include/aclgpu.h: acl_check_one_submit /
// acl_check_completions).  A goroutine that calls CheckPermission (reference pkg/authz/check.go:48 through the errgroup of
// check.go:76-94, watch.go:50) registers a channel under a fresh tag, submits, and parks on the channel -- a user-space
// switch.  ONE poller goroutine, pinned to its OS thread, drains the engine's completion queue and hands each answer to
// its channel.  With acl_check_one every check paid a futex sleep and a futex wake-up in the kernel (~17 us per check on
// the measured hosts: < 1 M checks/s on 16 cores whatever the device did); here a wake-up is paid per device pass
// (profiles/r02_batcher_ab.txt: 1.6 / 2.0 / 4.2 M checks/s at 64 / 256 / 1 024 callers from a C++ harness of the same shape).
type completion struct {
	rc, err int32
	perm    uint8
}

type lookupCompletion struct {
	rc  int32
	msg string       // why a failed lookup failed, where the shim knows (the completion record carries a code only)
	bm  []C.uint32_t // the result row, copied out of the engine's allocation
}

type completions struct {
	mu       sync.Mutex
	waiting  map[uint64]chan completion
	lwaiting map[uint64]chan lookupCompletion
	next     uint64
	closed   int32
	done     chan struct{}
	ldone    chan struct{}
}

func (e *Engine) startPoller() {
	e.cq = &completions{waiting: make(map[uint64]chan completion), lwaiting: make(map[uint64]chan lookupCompletion), done: make(chan struct{}), ldone: make(chan struct{})}
	go e.pollLookups()
	go func() {
		runtime.LockOSThread() // blocks in C between passes: keep it off the scheduler's shared threads
		defer close(e.cq.done)
		buf := make([]C.acl_completion_t, 256)
		for atomic.LoadInt32(&e.cq.closed) == 0 {
			var n C.size_t
			if rc := C.acl_check_completions(e.h, &buf[0], C.size_t(len(buf)), 100_000_000 /* ns: notices Close */, &n); rc != 0 {
				return
			}
			if n == 0 {
				continue
			}
			e.cq.mu.Lock()
			for i := 0; i < int(n); i++ {
				tag := uint64(buf[i].tag)
				if ch, ok := e.cq.waiting[tag]; ok { // (absent: its caller's ctx ended first)
					delete(e.cq.waiting, tag)
					ch <- completion{int32(buf[i].rc), int32(buf[i].err), uint8(buf[i].perm)} // buffered: never blocks the poller
				}
			}
			e.cq.mu.Unlock()
		}
	}()
}

// pollLookups drains acl_lookup_completions: every finished LookupResources arrives as an engine-allocated row that is copied into Go memory
// and released (acl_free) here -- also the rows of requests whose caller's ctx ended first.
func (e *Engine) pollLookups() {
	runtime.LockOSThread()
	defer close(e.cq.ldone)
	// whoever is still parked in lookupOne when this poller ends -- the engine closing, or the completion queue failing -- gets an
	// Unavailable completion instead of waiting for ever on a ctx that may have no deadline (ADVICE r3)
	defer func() {
		e.cq.mu.Lock()
		for tag, ch := range e.cq.lwaiting {
			delete(e.cq.lwaiting, tag)
			ch <- lookupCompletion{rc: int32(codes.Unavailable), msg: "the engine's lookup completion queue has shut down"} // buffered
		}
		e.cq.mu.Unlock()
	}()
	buf := make([]C.acl_lookup_completion_t, 64)
	for atomic.LoadInt32(&e.cq.closed) == 0 {
		var n C.size_t
		if rc := C.acl_lookup_completions(e.h, &buf[0], C.size_t(len(buf)), 100_000_000 /* ns: notices Close */, &n); rc != 0 {
			return
		}
		for i := 0; i < int(n); i++ {
			c := lookupCompletion{rc: int32(buf[i].rc)}
			if buf[i].bitmap != nil {
				c.bm = make([]C.uint32_t, int(buf[i].words))
				copy(c.bm, unsafe.Slice(buf[i].bitmap, int(buf[i].words)))
				C.acl_free(unsafe.Pointer(buf[i].bitmap))
			}
			tag := uint64(buf[i].tag)
			e.cq.mu.Lock()
			ch, ok := e.cq.lwaiting[tag]
			delete(e.cq.lwaiting, tag)
			e.cq.mu.Unlock()
			if ok {
				ch <- c // buffered: never blocks the poller
			}
		}
	}
}

// lookupOne submits one LookupResources and waits for its row or for ctx (the HTTP request's: responsefilterer.go:165-170).
func (e *Engine) lookupOne(ctx context.Context, rt, perm, st, sid, srel *C.char) (lookupCompletion, error) {
	ch := make(chan lookupCompletion, 1)
	e.cq.mu.Lock()
	e.cq.next++
	tag := e.cq.next
	e.cq.lwaiting[tag] = ch
	e.cq.mu.Unlock()
	if rc := C.acl_lookup_one_submit(e.h, rt, perm, st, sid, srel, C.uint64_t(tag)); rc != 0 { // (the strings are interned before it returns)
		e.cq.mu.Lock()
		delete(e.cq.lwaiting, tag)
		e.cq.mu.Unlock()
		return lookupCompletion{}, lastError(rc)
	}
	select {
	case c := <-ch:
		return c, nil
	case <-ctx.Done():
		e.cq.mu.Lock()
		delete(e.cq.lwaiting, tag) // the poller frees the row when it arrives
		e.cq.mu.Unlock()
		return lookupCompletion{}, ctxError(ctx)
	}
}

func (e *Engine) stopPoller() {
	if e.cq == nil {
		return
	}
	atomic.StoreInt32(&e.cq.closed, 1)
	<-e.cq.done
	<-e.cq.ldone
}

// checkOne submits one check and waits for its answer or for ctx (responsefilterer.go:168: the HTTP request's ctx ends a
// waiting check; its completion is then dropped by the poller).
func (e *Engine) checkOne(ctx context.Context, it *C.acl_check_item_t) (completion, error) {
	ch := make(chan completion, 1)
	e.cq.mu.Lock()
	e.cq.next++
	tag := e.cq.next
	e.cq.waiting[tag] = ch
	e.cq.mu.Unlock()
	if rc := C.acl_check_one_submit(e.h, it, C.uint64_t(tag)); rc != 0 { // copies what it needs: `it` may be freed on return
		e.cq.mu.Lock()
		delete(e.cq.waiting, tag)
		e.cq.mu.Unlock()
		return completion{}, lastError(rc)
	}
	select {
	case c := <-ch:
		return c, nil
	case <-ctx.Done():
		e.cq.mu.Lock()
		delete(e.cq.waiting, tag)
		e.cq.mu.Unlock()
		return completion{}, ctxError(ctx)
	}
}

// ctxError: what a gRPC client returns when its context ends (responsefilterer.go:170 looks for codes.Canceled).
func ctxError(ctx context.Context) error {
	if ctx.Err() == context.DeadlineExceeded {
		return status.Error(codes.DeadlineExceeded, ctx.Err().Error())
	}
	return status.Error(codes.Canceled, ctx.Err().Error())
}
