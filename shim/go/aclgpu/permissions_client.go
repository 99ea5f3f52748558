package aclgpu

/*
#include "shim.h"
*/
import "C"

import (
	"context"
	"io"
	"math/bits"
	"runtime/cgo"
	"sync/atomic"
	"time"
	"unsafe"

	v1 "github.com/authzed/authzed-go/proto/authzed/api/v1"
	"google.golang.org/grpc"
	"google.golang.org/grpc/codes"
	"google.golang.org/grpc/metadata"
	"google.golang.org/grpc/status"
)

type permissionsClient struct{ e *Engine }

// NewPermissionsClient is what goes into proxy.Options.PermissionsClient (pkg/proxy/options.go:82).
func NewPermissionsClient(e *Engine) v1.PermissionsServiceClient { return &permissionsClient{e} }

// CheckPermission: pkg/authz/watch.go:50 and every 1-item check expression (check.go:23-48).  Concurrent callers share
// one device pass through the micro-batcher.  With a batcher running the goroutine parks on a channel and the engine's
// completion queue answers it (completions.go: no OS thread blocked in C per check); without one it falls back to the
// blocking acl_check_one_opts (a device pass of its own).
func (p *permissionsClient) CheckPermission(ctx context.Context, in *v1.CheckPermissionRequest, _ ...grpc.CallOption) (*v1.CheckPermissionResponse, error) {
	var cs cstrings
	defer cs.free()
	it := cs.item(in.Resource, in.Permission, in.Subject)
	var perm C.uint8_t
	var perr C.int32_t
	if p.e.cq != nil {
		c, err := p.e.checkOne(ctx, &it)
		if err != nil {
			return nil, err
		}
		if c.rc != 0 {
			return nil, status.Error(codes.Code(c.rc), "the device pass that carried the check failed ("+codes.Code(c.rc).String()+"): "+describeCheck(in.Resource, in.Permission, in.Subject))
		}
		perm, perr = C.uint8_t(c.perm), C.int32_t(c.err)
	} else {
		opts, stop := callOpts(ctx) // ctx cancellation / deadline reach the engine through acl_call_opts_t
		defer stop()
		if rc := C.acl_check_one_opts(p.e.h, &it, &perm, &perr, opts); rc != 0 {
			return nil, lastError(rc)
		}
	}
	if perr != 0 {
		return nil, status.Error(itemCode(perr), itemMessage(perr, in.Resource, in.Permission, in.Subject)) // e.g. InvalidArgument for an empty request
	}
	return &v1.CheckPermissionResponse{CheckedAt: p.e.zedToken(), Permissionship: v1.CheckPermissionResponse_Permissionship(perm)}, nil
}

// CheckBulkPermissions: pkg/authz/check.go:48, postfilter.go:134.  Pairs[i] answers Items[i] (check.go:54-57).
func (p *permissionsClient) CheckBulkPermissions(ctx context.Context, in *v1.CheckBulkPermissionsRequest, _ ...grpc.CallOption) (*v1.CheckBulkPermissionsResponse, error) {
	n := len(in.Items)
	out := &v1.CheckBulkPermissionsResponse{CheckedAt: p.e.zedToken(), Pairs: make([]*v1.CheckBulkPermissionsPair, n)}
	if n == 0 {
		return out, nil
	}
	// A PACKED request (acl_check_bulk_packed, engine.go packedItems): every DISTINCT string of the request once in one C buffer and six dictionary
	// indices per item -- three allocations per call instead of six C.CString mallocs per item; the template's type / permission names and the
	// user every pair of a PostFilter call names (postfilter.go:88-119) are one dictionary entry each, equal by index.
	pi := newPackedItems(n)
	defer pi.free()
	for i, it := range in.Items {
		pi.set(i, it.Resource, it.Permission, it.Subject)
	}
	perm := make([]C.uint8_t, n)
	errs := make([]C.int32_t, n)
	opts, stop := callOpts(ctx) // ctx cancellation / deadline reach the engine through acl_call_opts_t (check.go:48 passes the request's ctx)
	defer stop()
	if rc := C.acl_check_bulk_packed(p.e.h, &pi.req, &perm[0], &errs[0], opts); rc != 0 {
		return nil, lastError(rc) // (a request the API's validation refuses: the message names the item, the field, the value and the pattern)
	}
	for i := range in.Items {
		pair := &v1.CheckBulkPermissionsPair{Request: in.Items[i]}
		if errs[i] != 0 {
			pair.Response = &v1.CheckBulkPermissionsPair_Error{Error: status.New(itemCode(errs[i]), itemMessage(errs[i], in.Items[i].Resource, in.Items[i].Permission, in.Items[i].Subject)).Proto()}
		} else {
			pair.Response = &v1.CheckBulkPermissionsPair_Item{Item: &v1.CheckBulkPermissionsResponseItem{
				Permissionship: v1.CheckPermissionResponse_Permissionship(perm[i])}} // ACL_PERM_* == the proto enum values
		}
		out.Pairs[i] = pair
	}
	return out, nil
}

// LookupResources: pkg/authz/lookups.go:65; the proxy drains the stream until io.EOF (lookups.go:75-83).
func (p *permissionsClient) LookupResources(ctx context.Context, in *v1.LookupResourcesRequest, _ ...grpc.CallOption) (v1.PermissionsService_LookupResourcesClient, error) {
	var cs cstrings
	defer cs.free()
	var st, sid, srel string
	if in.Subject != nil && in.Subject.Object != nil {
		st, sid, srel = in.Subject.Object.ObjectType, in.Subject.Object.ObjectId, in.Subject.OptionalRelation
	}
	rt := cs.add(in.ResourceObjectType)
	typeID := C.acl_type_id(p.e.h, rt)
	// The result bitmap is engine-owned (acl_lookup_resources_alloc): it is sized when the walk runs, so objects interned by a
	// racing WriteRelationships can never make a caller-sized buffer "too small".  Concurrent list requests of the same
	// (type, permission, subject class) share one batched reverse walk (the micro-batcher); the HTTP request's ctx
	// (responsefilterer.go:165-170) cancels the call while it queues and between level bursts of the walk.
	if p.e.cq != nil {
		// no OS thread blocked in C per prefilter (responsefilterer.go:165-183 starts one goroutine per list request): submit, park on a
		// channel, the lookup poller hands the row over (completions.go)
		c, err := p.e.lookupOne(ctx, rt, cs.add(in.Permission), cs.add(st), cs.add(sid), cs.add(srel))
		if err != nil {
			return nil, err
		}
		if c.rc != 0 {
			msg := c.msg
			if msg == "" {
				msg = "LookupResources failed in the engine (code " + codes.Code(c.rc).String() + "); ACL_TRACE=1 logs the reason where the completion is produced"
			}
			// (a candidate whose forward Check erred fails the call with the ITEM's code, ACL_ERR_DEPTH: SpiceDB's "max depth exceeded" is a
			// ResourceExhausted; the reference ends the stream on it and fails the list request, lookups.go:75-83, responsefilterer.go:196-204)
			return nil, status.Error(itemCode(C.int32_t(c.rc)), msg)
		}
		return &bitmapStream{ctx: ctx, e: p.e, typeID: typeID, bm: c.bm, at: p.e.zedToken()}, nil
	}
	var bmp *C.uint32_t
	var words C.size_t
	var count C.uint64_t
	opts, stop := callOpts(ctx)
	defer stop()
	if rc := C.acl_lookup_resources_alloc(p.e.h, rt, cs.add(in.Permission), cs.add(st), cs.add(sid), cs.add(srel), opts, &bmp, &words, &count); rc != 0 {
		return nil, status.Error(itemCode(C.int32_t(rc)), C.GoString(C.acl_last_error()))
	}
	bm := make([]C.uint32_t, int(words))
	copy(bm, unsafe.Slice(bmp, int(words)))
	C.acl_free(unsafe.Pointer(bmp))
	return &bitmapStream{ctx: ctx, e: p.e, typeID: typeID, bm: bm, at: p.e.zedToken()}, nil
}

// bitmapStream turns the result bitmap into the stream the reference consumes: one HAS_PERMISSION message per set bit.  The names come from the engine a BLOCK at a
// time (acl_bitmap_names: one cgo call and one turn at the engine's names lock per 512 results, not per result -- a list of 10 000 allowed pods was 10 000 calls);
// they are copied under that lock, so an id that is given to another object later cannot change what was streamed.
type bitmapStream struct {
	ctx    context.Context
	e      *Engine
	typeID C.int
	bm     []C.uint32_t
	cursor C.uint64_t // first bit not fetched yet
	buf    []C.char   // the current block's names, back to back
	ends   []C.uint32_t
	n, k   int // names in the block, next one to hand out
	at     *v1.ZedToken
}

func (s *bitmapStream) Recv() (*v1.LookupResourcesResponse, error) {
	if err := s.ctx.Err(); err != nil { // LR uses the HTTP request context (responsefilterer.go:168)
		return nil, status.FromContextError(err).Err()
	}
	if s.k == s.n {
		if len(s.bm) == 0 {
			return nil, io.EOF
		}
		if s.buf == nil {
			s.buf = make([]C.char, 64<<10)
			s.ends = make([]C.uint32_t, 512)
		}
		var n C.size_t
		if rc := C.acl_bitmap_names(s.e.h, s.typeID, &s.bm[0], C.size_t(len(s.bm)), &s.cursor, &s.buf[0], C.size_t(len(s.buf)), &s.ends[0], C.size_t(len(s.ends)), &n); rc != 0 {
			return nil, status.Error(codes.Code(rc), C.GoString(C.acl_last_error()))
		}
		if n == 0 {
			return nil, io.EOF
		}
		s.n, s.k = int(n), 0
	}
	from := 0
	if s.k > 0 {
		from = int(s.ends[s.k-1])
	}
	name := "" // (an id that lost its name meanwhile: an empty id, as before)
	if end := int(s.ends[s.k]); end > from {
		name = C.GoStringN(&s.buf[from], C.int(end-from))
	}
	s.k++
	return &v1.LookupResourcesResponse{LookedUpAt: s.at, ResourceObjectId: name,
		Permissionship: v1.LookupPermissionship_LOOKUP_PERMISSIONSHIP_HAS_PERMISSION}, nil
}
func (s *bitmapStream) Header() (metadata.MD, error) { return nil, nil }
func (s *bitmapStream) Trailer() metadata.MD         { return nil }
func (s *bitmapStream) CloseSend() error             { return nil }
func (s *bitmapStream) Context() context.Context     { return s.ctx }
func (s *bitmapStream) SendMsg(any) error            { return nil }
func (s *bitmapStream) RecvMsg(any) error            { return io.EOF }

// WriteRelationships: pkg/authz/distributedtx/activity.go:60.  Atomic; preconditions see the pre-write state; AlreadyExists
// for a duplicate CREATE and FailedPrecondition for an unmet precondition drive the workflow's conflict logic
// (workflow.go:187-201); InvalidArgument is unrecoverable during rollback (workflow.go:115-119).
func (p *permissionsClient) WriteRelationships(ctx context.Context, in *v1.WriteRelationshipsRequest, _ ...grpc.CallOption) (*v1.WriteRelationshipsResponse, error) {
	var cs cstrings
	defer cs.free()
	ups := make([]C.acl_update_t, len(in.Updates)+1)
	for i, u := range in.Updates {
		if u == nil || u.Relationship == nil || u.Relationship.Resource == nil || u.Relationship.Subject == nil || u.Relationship.Subject.Object == nil {
			return nil, status.Error(codes.InvalidArgument, "invalid WriteRelationshipsRequest: update without relationship, resource or subject")
		}
		r := u.Relationship
		rel := C.acl_relationship_t{resource_type: cs.add(r.Resource.ObjectType), resource_id: cs.add(r.Resource.ObjectId), relation: cs.add(r.Relation),
			subject_type: cs.add(r.Subject.Object.ObjectType), subject_id: cs.add(r.Subject.Object.ObjectId), subject_relation: cs.add(r.Subject.OptionalRelation)}
		if r.OptionalExpiresAt != nil {
			rel.expires_at = C.int64_t(r.OptionalExpiresAt.Seconds)
		}
		ups[i] = C.acl_update_t{op: C.int32_t(u.Operation), rel: rel} // OPERATION_CREATE/TOUCH/DELETE == ACL_OP_*
	}
	pre := make([]C.acl_filter_t, len(in.OptionalPreconditions)+1)
	for i, pc := range in.OptionalPreconditions {
		if pc == nil || pc.Filter == nil {
			return nil, status.Error(codes.InvalidArgument, "invalid WriteRelationshipsRequest: precondition without filter")
		}
		pre[i] = cs.filter(pc.Filter, C.int32_t(pc.Operation)) // OPERATION_MUST_NOT_MATCH/MUST_MATCH == ACL_PRE_*
	}
	var rev C.uint64_t
	if rc := C.acl_write(p.e.h, &ups[0], C.int(len(in.Updates)), &pre[0], C.int(len(in.OptionalPreconditions)), &rev); rc != 0 {
		return nil, lastError(rc)
	}
	return &v1.WriteRelationshipsResponse{WrittenAt: p.e.zedToken()}, nil
}

// DeleteRelationships: e2e/util_test.go:66.  OptionalPreconditions are evaluated against the pre-delete state, atomically
// with the delete (acl_delete_by_filter_pre); a limit / partial deletion is not something the engine offers: refused.
func (p *permissionsClient) DeleteRelationships(ctx context.Context, in *v1.DeleteRelationshipsRequest, _ ...grpc.CallOption) (*v1.DeleteRelationshipsResponse, error) {
	if in.RelationshipFilter == nil {
		return nil, status.Error(codes.InvalidArgument, "invalid DeleteRelationshipsRequest: no filter")
	}
	if in.OptionalLimit != 0 || in.OptionalAllowPartialDeletions {
		return nil, status.Error(codes.Unimplemented, "DeleteRelationships with a limit / partial deletions is not implemented by the GPU ACL engine")
	}
	var cs cstrings
	defer cs.free()
	f := cs.filter(in.RelationshipFilter, 0)
	pre := make([]C.acl_filter_t, len(in.OptionalPreconditions)+1)
	for i, pc := range in.OptionalPreconditions {
		if pc == nil || pc.Filter == nil {
			return nil, status.Error(codes.InvalidArgument, "invalid DeleteRelationshipsRequest: precondition without filter")
		}
		pre[i] = cs.filter(pc.Filter, C.int32_t(pc.Operation))
	}
	var n, rev C.uint64_t
	if rc := C.acl_delete_by_filter_pre(p.e.h, &f, &pre[0], C.int(len(in.OptionalPreconditions)), &n, &rev); rc != 0 {
		return nil, lastError(rc)
	}
	return &v1.DeleteRelationshipsResponse{DeletedAt: p.e.zedToken(), RelationshipsDeletedCount: uint64(n)}, nil
}

// ReadRelationships: activity.go:107,154; e2e/util_test.go:27.  The matches are buffered, then streamed.
func (p *permissionsClient) ReadRelationships(ctx context.Context, in *v1.ReadRelationshipsRequest, _ ...grpc.CallOption) (v1.PermissionsService_ReadRelationshipsClient, error) {
	var cs cstrings
	defer cs.free()
	f := cs.filter(in.RelationshipFilter, 0)
	sink := &readSink{}
	hdl := cgo.NewHandle(sink)
	defer hdl.Delete()
	if rc := C.acl_read_go(p.e.h, &f, unsafe.Pointer(uintptr(hdl))); rc != 0 {
		return nil, lastError(rc)
	}
	return &relStream{ctx: ctx, rels: sink.rels, at: p.e.zedToken()}, nil
}

type relStream struct {
	ctx  context.Context
	rels []*v1.Relationship
	at   *v1.ZedToken
}

func (s *relStream) Recv() (*v1.ReadRelationshipsResponse, error) {
	if len(s.rels) == 0 {
		return nil, io.EOF
	}
	r := s.rels[0]
	s.rels = s.rels[1:]
	return &v1.ReadRelationshipsResponse{ReadAt: s.at, Relationship: r}, nil
}
func (s *relStream) Header() (metadata.MD, error) { return nil, nil }
func (s *relStream) Trailer() metadata.MD         { return nil }
func (s *relStream) CloseSend() error             { return nil }
func (s *relStream) Context() context.Context     { return s.ctx }
func (s *relStream) SendMsg(any) error            { return nil }
func (s *relStream) RecvMsg(any) error            { return io.EOF }

// Never called by the proxy (SURVEY.md 8(b)): Unimplemented, as a real server without the feature would answer.
func (p *permissionsClient) LookupSubjects(context.Context, *v1.LookupSubjectsRequest, ...grpc.CallOption) (v1.PermissionsService_LookupSubjectsClient, error) {
	return nil, status.Error(codes.Unimplemented, "not implemented by the GPU ACL engine")
}
func (p *permissionsClient) ExpandPermissionTree(context.Context, *v1.ExpandPermissionTreeRequest, ...grpc.CallOption) (*v1.ExpandPermissionTreeResponse, error) {
	return nil, status.Error(codes.Unimplemented, "not implemented by the GPU ACL engine")
}
func (p *permissionsClient) ExportBulkRelationships(context.Context, *v1.ExportBulkRelationshipsRequest, ...grpc.CallOption) (v1.PermissionsService_ExportBulkRelationshipsClient, error) {
	return nil, status.Error(codes.Unimplemented, "not implemented by the GPU ACL engine")
}
func (p *permissionsClient) ImportBulkRelationships(context.Context, ...grpc.CallOption) (v1.PermissionsService_ImportBulkRelationshipsClient, error) {
	return nil, status.Error(codes.Unimplemented, "not implemented by the GPU ACL engine")
}

// KeepMask is the fused form of filterItemsWithBulkPermissions (pkg/authz/postfilter.go:58-182) for callers patched to use
// it: pairs = the resolved PostFilter checks of all list items, off[i]..off[i+1] = item i's pairs; true = keep the item.
// The pairs go over PACKED (acl_check_bulk_keep_packed): the reference's PostFilter names ONE subject for all of them
// (postfilter.go:88-119), which the engine recognises by dictionary index and answers with one reverse walk + bit tests
// instead of K forward walks; any other call takes the forward path behind the same entry point, with the same mask.
func (e *Engine) KeepMask(pairs []*v1.CheckBulkPermissionsRequestItem, off []uint32) ([]bool, error) {
	k := len(off) - 1
	if k <= 0 {
		return nil, nil
	}
	keep := make([]C.uint8_t, k)
	if len(pairs) > 0 {
		pi := newPackedItems(len(pairs))
		defer pi.free()
		for i, it := range pairs {
			pi.set(i, it.Resource, it.Permission, it.Subject)
		}
		if rc := C.acl_check_bulk_keep_packed(e.h, &pi.req, (*C.uint32_t)(unsafe.Pointer(&off[0])), C.size_t(k), &keep[0]); rc != 0 {
			return nil, lastError(rc)
		}
	} else {
		for i := range keep {
			keep[i] = 1 // (an item without pairs is kept: postfilter.go:145-150)
		}
	}
	out := make([]bool, k)
	for i := range out {
		out[i] = keep[i] != 0
	}
	return out, nil
}

// itemCode maps a per-item error of the engine to the gRPC code the pair carries: the engine's codes ARE gRPC codes except
// ACL_ERR_DEPTH (100), SpiceDB's "max depth exceeded" -- a ResourceExhausted there.
// itemMessage says what a pair's error is ABOUT: the reference denies on any error (pkg/authz/check.go:55-60) and logs it, so "check failed" left an
// operator with nothing (VERDICT r5 next #8).  The engine's own message for a refused REQUEST is acl_last_error(); a pair only carries a code.
func itemMessage(c C.int32_t, resource *v1.ObjectReference, permission string, subject *v1.SubjectReference) string {
	what := describeCheck(resource, permission, subject)
	switch c {
	case C.ACL_ERR_INVALID_ARGUMENT:
		return "invalid request: a member of " + what + " is empty or does not match the API's pattern (object ids: ^[a-zA-Z0-9/_|\\-=+]{1,1024}$, names: ^[a-z][a-z0-9_]{1,62}[a-z0-9]$)"
	case C.ACL_ERR_FAILED_PRECONDITION:
		return "object definition, relation or permission not found in the schema: " + what
	case C.ACL_ERR_DEPTH:
		return "max depth exceeded (a recursive or too deep data dependency) while checking " + what
	}
	return "check of " + what + " failed"
}
func describeCheck(resource *v1.ObjectReference, permission string, subject *v1.SubjectReference) string {
	r, s := "<no resource>", "<no subject>"
	if resource != nil {
		r = resource.ObjectType + ":" + resource.ObjectId
	}
	if subject != nil && subject.Object != nil {
		s = subject.Object.ObjectType + ":" + subject.Object.ObjectId
		if subject.OptionalRelation != "" {
			s += "#" + subject.OptionalRelation
		}
	}
	return r + "#" + permission + "@" + s
}

func itemCode(c C.int32_t) codes.Code {
	if c == C.ACL_ERR_DEPTH {
		return codes.ResourceExhausted
	}
	return codes.Code(c)
}

// callOpts turns a context into acl_call_opts_t: a deadline becomes timeout_ns, cancellation raises an int32 flag the engine
// polls (while the call waits for a device context or the micro-batcher, and between level bursts of a walk).  The flag
// lives in C memory: the engine reads it while the goroutine is blocked in cgo.  stop() releases it.
func callOpts(ctx context.Context) (*C.acl_call_opts_t, func()) {
	done := ctx.Done()
	dl, hasDL := ctx.Deadline()
	if done == nil && !hasDL {
		return nil, func() {}
	}
	o := (*C.acl_call_opts_t)(C.calloc(1, C.size_t(unsafe.Sizeof(C.acl_call_opts_t{}))))
	flag := (*C.int32_t)(C.calloc(1, 4))
	o.cancel = flag
	if hasDL {
		if d := time.Until(dl); d > 0 {
			o.timeout_ns = C.int64_t(d.Nanoseconds())
		} else {
			o.timeout_ns = 1
		}
	}
	quit := make(chan struct{})
	finished := make(chan struct{})
	go func() {
		defer close(finished)
		select {
		case <-done:
			atomic.StoreInt32((*int32)(unsafe.Pointer(flag)), 1)
		case <-quit:
		}
	}()
	return o, func() {
		close(quit)
		<-finished
		C.free(unsafe.Pointer(flag))
		C.free(unsafe.Pointer(o))
	}
}

// FilterListResponse is filterListResponse (pkg/authz/postfilter.go:17-55) on the list response's bytes for rules whose
// PostFilter templates are plain placeholders ({{name}}, {{namespace}}, {{namespacedName}}, {{user.name}} -- the form of
// deploy/rules.yaml:68): one scan of the body, one device pass for all K x F pairs, the original bytes with the dropped
// items cut out.  Rules that need the full Bloblang / CEL environment keep resolving in Go and call KeepMask.
func (e *Engine) FilterListResponse(body []byte, templates []string, userName string) ([]byte, error) {
	return e.FilterListResponseReq(body, templates, userName, "", "", "")
}

// FilterListResponseReq also hands over the kube request the list answers (input.Request: RequestInfo.Name / .Namespace / .Resource), so that
// every item's placeholders resolve as rules.NewResolveInput resolves them (pkg/rules/rules.go:315-342): the item's own metadata first, the
// request's name / namespace where the item has none, and no namespace for the `namespaces` resource.
func (e *Engine) FilterListResponseReq(body []byte, templates []string, userName, reqName, reqNamespace, reqResource string) ([]byte, error) {
	if len(body) == 0 {
		// the reference fails to parse an empty body (postfilter.go:21-24): so does the engine -- let it say so
		body = []byte{}
	}
	var cs cstrings
	defer cs.free()
	tp := make([]*C.char, len(templates)+1)
	for i, t := range templates {
		tp[i] = cs.add(t)
	}
	req := (*C.acl_list_request_t)(C.calloc(1, C.size_t(unsafe.Sizeof(C.acl_list_request_t{}))))
	defer C.free(unsafe.Pointer(req))
	req.name, req.namespace_, req.resource = cs.add(reqName), cs.add(reqNamespace), cs.add(reqResource)
	var bp *C.char
	if len(body) > 0 {
		bp = (*C.char)(unsafe.Pointer(&body[0]))
	} else {
		bp = cs.add("")
	}
	var out *C.char
	var n C.size_t
	if rc := C.acl_filter_list_response_req(e.h, bp, C.size_t(len(body)), &tp[0], C.size_t(len(templates)), cs.add(userName), req, &out, &n, nil, nil); rc != 0 {
		return nil, lastError(rc)
	}
	defer C.acl_free(unsafe.Pointer(out))
	return C.GoBytes(unsafe.Pointer(out), C.int(n)), nil
}

// Allowed is a LookupResources result kept as the engine produced it: the allowed-id bitmap of one (resource type, permission, subject).  The
// reference drains the stream into a set of NamespacedNames (pkg/authz/lookups.go:65-131); a caller that only wants the kube response filtered
// keeps the bitmap and hands the response's bytes to FilterResponse.
type Allowed struct {
	e      *Engine
	typeID C.int
	bm     []C.uint32_t
}

// LookupAllowed runs the LookupResources of a PreFilter (lookups.go:49-83) and keeps its result as a bitmap.
func (e *Engine) LookupAllowed(ctx context.Context, in *v1.LookupResourcesRequest) (*Allowed, error) {
	st, err := (&permissionsClient{e: e}).LookupResources(ctx, in)
	if err != nil {
		return nil, err
	}
	bs := st.(*bitmapStream)
	return &Allowed{e: e, typeID: bs.typeID, bm: bs.bm}, nil
}

// BodyKind says what the kube response holds: a list ("items"), a Table ("rows", Accept: ...;as=Table) or one object.
type BodyKind int

const (
	BodyList BodyKind = iota
	BodyTable
	BodyObject
)

// FilterResponse is filterList / filterTable / filterObject (pkg/authz/responsefilterer.go:349-416) on the response's bytes: the elements
// whose (namespace, name) the PreFilter admits stay, the others are cut out of the original bytes; a single object outside the set is the
// reference's "unauthorized" (codes.PermissionDenied here; writeResp turns it into a 401, responsefilterer.go:716-727).  idTemplate is the
// rule's fromObjectID mapping read forwards: "{{name}}" for `fromObjectIDNameExpr: {{resourceId}}` (deploy/rules.yaml:51),
// "{{namespacedName}}" for split_namespace / split_name (rules.yaml:105-106).  Rules with other expressions keep using the stream.
func (a *Allowed) FilterResponse(body []byte, kind BodyKind, idTemplate string) ([]byte, error) {
	var cs cstrings
	defer cs.free()
	var bp *C.char
	if len(body) > 0 {
		bp = (*C.char)(unsafe.Pointer(&body[0]))
	} else {
		bp = cs.add("")
	}
	var bm *C.uint32_t
	if len(a.bm) > 0 {
		bm = &a.bm[0]
	}
	var out *C.char
	var n C.size_t
	if rc := C.acl_prefilter_response(a.e.h, a.typeID, bm, C.size_t(len(a.bm)), cs.add(idTemplate), C.int(kind), bp, C.size_t(len(body)), &out, &n, nil, nil); rc != 0 {
		return nil, lastError(rc)
	}
	defer C.acl_free(unsafe.Pointer(out))
	return C.GoBytes(unsafe.Pointer(out), C.int(n)), nil
}
