// Package aclgpu binds libaclgpu.so (include/aclgpu.h), the MI355X-native batched ACL-check engine, and implements the
// authzed v1.PermissionsServiceClient / v1.WatchServiceClient interfaces the proxy holds in proxy.Options
// (reference pkg/proxy/options.go:81-82).  UNBUILT in this repository: see README.md.
package aclgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../spicedb-kubeapi-proxy_amd/lib -laclgpu -Wl,-rpath,${SRCDIR}/../../../spicedb-kubeapi-proxy_amd/lib
#include "shim.h"
*/
import "C"

import (
	"os"
	"sort"
	"strconv"
	"unsafe"

	v1 "github.com/authzed/authzed-go/proto/authzed/api/v1"
	"google.golang.org/grpc/codes"
	"google.golang.org/grpc/status"
)

// Config mirrors acl_config_t plus the micro-batcher settings (acl_batcher_start).
type Config struct {
	Device             int32  // HIP device ordinal, -1 = current
	Devices            []int32 // several ordinals: ONE engine in front of one HBM snapshot per device (acl_open_replicas); overrides Device
	FrontierEntries    uint64 // 0 = default
	BatchMaxItems      uint32 // 0 = no micro-batching of single checks
	BatchMaxWaitMicros uint32
	Contexts           uint32 // evaluations in flight on the device at once (own HIP stream each); 0 = default (4)
}

// Engine owns one acl_engine_t.  All methods are safe for concurrent use (the C ABI is).
type Engine struct {
	h  *C.acl_engine_t
	cq *completions // non-nil while the micro-batcher runs: single checks go through the completion queue (completions.go)
}

func lastError(rc C.int) error {
	return status.Error(codes.Code(rc), C.GoString(C.acl_last_error())) // return codes ARE gRPC codes (aclgpu.h)
}

// Open replaces spicedb.NewServer (reference pkg/spicedb/spicedb.go:18-71): schema + bootstrap relationships
// (pkg/spicedb/bootstrap.yaml) -> a running engine.
func Open(cfg Config, schema, relationships string) (*Engine, error) {
	var h *C.acl_engine_t
	c := C.acl_config_t{device: C.int32_t(cfg.Device), frontier_entries: C.uint64_t(cfg.FrontierEntries), contexts: C.uint32_t(cfg.Contexts)}
	if len(cfg.Devices) > 0 {
		// the proxy holds ONE PermissionsClient (options.go:371-377) and its dual-write worker shares it (server.go:136-153): every write is
		// patched into every replica before its next read, calls are spread over the devices by load
		devs := make([]C.int32_t, len(cfg.Devices))
		for i, d := range cfg.Devices {
			devs[i] = C.int32_t(d)
		}
		if rc := C.acl_open_replicas(&c, &devs[0], C.uint32_t(len(devs)), &h); rc != 0 {
			return nil, lastError(rc)
		}
	} else if rc := C.acl_open(&c, &h); rc != 0 {
		return nil, lastError(rc)
	}
	cs, cr := C.CString(schema), C.CString(relationships)
	defer C.free(unsafe.Pointer(cs))
	defer C.free(unsafe.Pointer(cr))
	if rc := C.acl_load_bootstrap(h, cs, C.size_t(len(schema)), cr, C.size_t(len(relationships))); rc != 0 {
		C.acl_close(h)
		return nil, lastError(rc)
	}
	if cfg.BatchMaxItems > 0 {
		if rc := C.acl_batcher_start(h, C.uint32_t(cfg.BatchMaxItems), C.uint32_t(cfg.BatchMaxWaitMicros)); rc != 0 {
			C.acl_close(h)
			return nil, lastError(rc)
		}
		e := &Engine{h: h}
		e.startPoller()
		return e, nil
	}
	return &Engine{h: h}, nil
}

// OpenBootstrap replaces spicedb.NewServer(ctx, bootstrapFilePath, bootstrapContent) (reference pkg/spicedb/spicedb.go:18-24) for
// `gpu://` endpoints (shim/patches/options_gpu_scheme.patch): the bootstrap comes from the content map when it has entries, else from the
// file the endpoint URL names, else from `defaultBootstrap` (the reference's embedded pkg/spicedb/bootstrap.yaml, handed in by the caller:
// the default stays in the reference binary).  Every source is YAML `{schema, relationships}`; the engine reads it itself
// (acl_load_bootstrap_yaml), several files are merged as `---` documents in name order.
func OpenBootstrap(cfg Config, bootstrapFilePath string, bootstrapContent map[string][]byte, defaultBootstrap []byte) (*Engine, error) {
	var doc []byte
	switch {
	case len(bootstrapContent) > 0:
		names := make([]string, 0, len(bootstrapContent))
		for n := range bootstrapContent {
			names = append(names, n)
		}
		sort.Strings(names)
		for i, n := range names {
			if i > 0 {
				doc = append(doc, []byte("\n---\n")...)
			}
			doc = append(doc, bootstrapContent[n]...)
		}
	case len(bootstrapFilePath) > 0:
		b, err := os.ReadFile(bootstrapFilePath)
		if err != nil {
			return nil, status.Errorf(codes.InvalidArgument, "bootstrap file %q: %v", bootstrapFilePath, err)
		}
		doc = b
	default:
		doc = defaultBootstrap
	}
	e, err := Open(cfg, "definition aclgpu_placeholder {}", "") // (an engine needs a schema to open its tables; the bootstrap replaces it)
	if err != nil {
		return nil, err
	}
	cd := C.CBytes(doc)
	defer C.free(cd)
	if rc := C.acl_load_bootstrap_yaml(e.h, (*C.char)(cd), C.size_t(len(doc))); rc != 0 {
		err := lastError(rc)
		e.Close()
		return nil, err
	}
	return e, nil
}

func (e *Engine) Close() {
	e.stopPoller()
	C.acl_close(e.h)
}

// zedToken: the store revision as an opaque token (activity.go:76 only stores and compares it).
func (e *Engine) zedToken() *v1.ZedToken {
	return &v1.ZedToken{Token: "aclgpu-" + strconv.FormatUint(uint64(C.acl_revision(e.h)), 10)}
}

// cstrings allocates C strings and frees them together.
type cstrings struct{ ptrs []unsafe.Pointer }

func (c *cstrings) add(s string) *C.char {
	p := C.CString(s)
	c.ptrs = append(c.ptrs, unsafe.Pointer(p))
	return p
}
func (c *cstrings) free() {
	for _, p := range c.ptrs {
		C.free(p)
	}
}

// packedItems builds an acl_packed_request_t (aclgpu.h): the request's DISTINCT strings once, back to back, and six dictionary indices per item.  The
// strings of a bulk request repeat -- one rule template per item (pkg/authz/check.go:23-39), one user for every pair of a PostFilter call
// (postfilter.go:88-119) -- so an item costs 24 bytes of indices instead of six {pointer, length} views, the engine finds the repeated fields equal BY
// INDEX, and resolves a name that many items carry once per call.  Two C allocations per call (the bytes + offsets, the indices).
type packedItems struct {
	n       int
	dict    map[string]uint32
	bytes   []byte
	offsets []uint32 // [strings + 1]
	idx     []uint32 // [n][6]
	cbytes  unsafe.Pointer
	coffs   unsafe.Pointer
	cidx    unsafe.Pointer
	req     C.acl_packed_request_t
}

const packedNone = ^uint32(0) // ACL_PACKED_NONE

func newPackedItems(n int) *packedItems {
	return &packedItems{n: n, dict: make(map[string]uint32), offsets: []uint32{0}, idx: make([]uint32, 6*n)}
}
func (v *packedItems) ref(s string) uint32 {
	k, ok := v.dict[s]
	if !ok {
		k = uint32(len(v.offsets) - 1)
		v.dict[s] = k
		v.bytes = append(v.bytes, s...)
		v.offsets = append(v.offsets, uint32(len(v.bytes)))
	}
	return k
}
func (v *packedItems) set(i int, resource *v1.ObjectReference, permission string, subject *v1.SubjectReference) {
	r := v.idx[6*i : 6*i+6] // nil members (an empty request) stay absent: that pair is answered InvalidArgument (options_test.go:101-102)
	for f := range r {
		r[f] = packedNone
	}
	if resource != nil {
		r[0], r[1] = v.ref(resource.ObjectType), v.ref(resource.ObjectId)
	}
	r[2] = v.ref(permission)
	if subject != nil && subject.Object != nil {
		r[3], r[4] = v.ref(subject.Object.ObjectType), v.ref(subject.Object.ObjectId)
		if subject.OptionalRelation != "" {
			r[5] = v.ref(subject.OptionalRelation)
		}
	}
	if i == v.n-1 {
		v.finish()
	}
}
func (v *packedItems) finish() {
	v.cbytes = C.malloc(C.size_t(len(v.bytes) + 1))
	copy(unsafe.Slice((*byte)(v.cbytes), len(v.bytes)), v.bytes)
	v.coffs = C.malloc(C.size_t(4 * len(v.offsets)))
	copy(unsafe.Slice((*uint32)(v.coffs), len(v.offsets)), v.offsets)
	v.cidx = C.malloc(C.size_t(4*len(v.idx) + 4))
	copy(unsafe.Slice((*uint32)(v.cidx), len(v.idx)), v.idx)
	v.req = C.acl_packed_request_t{bytes: (*C.char)(v.cbytes), offsets: (*C.uint32_t)(v.coffs), n_strings: C.uint32_t(len(v.offsets) - 1),
		items: (*C.uint32_t)(v.cidx), n_items: C.size_t(v.n)}
}
func (v *packedItems) free() {
	C.free(v.cbytes)
	C.free(v.coffs)
	C.free(v.cidx)
}

func (c *cstrings) item(resource *v1.ObjectReference, permission string, subject *v1.SubjectReference) C.acl_check_item_t {
	var it C.acl_check_item_t // nil members (an empty request) stay NULL: the engine answers InvalidArgument (options_test.go:101-102)
	if resource != nil {
		it.resource_type, it.resource_id = c.add(resource.ObjectType), c.add(resource.ObjectId)
	}
	it.permission = c.add(permission)
	if subject != nil && subject.Object != nil {
		it.subject_type, it.subject_id = c.add(subject.Object.ObjectType), c.add(subject.Object.ObjectId)
		it.subject_relation = c.add(subject.OptionalRelation)
	}
	return it
}

func (c *cstrings) filter(f *v1.RelationshipFilter, op C.int32_t) C.acl_filter_t {
	out := C.acl_filter_t{op: op, resource_type: c.add(f.ResourceType)}
	if f.OptionalResourceId != "" {
		out.resource_id = c.add(f.OptionalResourceId)
	}
	if f.OptionalRelation != "" {
		out.relation = c.add(f.OptionalRelation)
	}
	if sf := f.OptionalSubjectFilter; sf != nil {
		out.subject_type = c.add(sf.SubjectType)
		if sf.OptionalSubjectId != "" {
			out.subject_id = c.add(sf.OptionalSubjectId)
		}
		if sf.OptionalRelation != nil {
			out.subject_relation = c.add(sf.OptionalRelation.Relation) // "" = only relationships without a subject relation
		}
	}
	return out
}

func relationshipFromC(r *C.acl_relationship_t) *v1.Relationship {
	return &v1.Relationship{
		Resource: &v1.ObjectReference{ObjectType: C.GoString(r.resource_type), ObjectId: C.GoString(r.resource_id)},
		Relation: C.GoString(r.relation),
		Subject: &v1.SubjectReference{
			Object:           &v1.ObjectReference{ObjectType: C.GoString(r.subject_type), ObjectId: C.GoString(r.subject_id)},
			OptionalRelation: C.GoString(r.subject_relation),
		},
	}
}
