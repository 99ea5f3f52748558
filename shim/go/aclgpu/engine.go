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

// viewItems builds an acl_check_item_v_t array in C memory over ONE C blob holding every distinct string of the request.
type viewItems struct {
	items *C.acl_check_item_v_t
	n     int
	blob  []byte            // staged in Go, copied to C once (finish)
	at    map[string]int    // string -> offset in blob
	refs  [][6][2]int       // per item and field: {offset, length}; offset -1 = absent
	cblob unsafe.Pointer
}

func newViewItems(n int) *viewItems {
	return &viewItems{n: n, at: make(map[string]int), refs: make([][6][2]int, n)}
}
func (v *viewItems) ref(s string, present bool) [2]int {
	if !present {
		return [2]int{-1, 0}
	}
	off, ok := v.at[s]
	if !ok {
		off = len(v.blob)
		v.at[s] = off
		v.blob = append(v.blob, s...)
	}
	return [2]int{off, len(s)}
}
func (v *viewItems) set(i int, resource *v1.ObjectReference, permission string, subject *v1.SubjectReference) {
	r := &v.refs[i] // nil members (an empty request) stay absent: the engine answers InvalidArgument (options_test.go:101-102)
	for f := range r {
		r[f] = [2]int{-1, 0}
	}
	if resource != nil {
		r[0], r[1] = v.ref(resource.ObjectType, true), v.ref(resource.ObjectId, true)
	}
	r[2] = v.ref(permission, true)
	if subject != nil && subject.Object != nil {
		r[3], r[4] = v.ref(subject.Object.ObjectType, true), v.ref(subject.Object.ObjectId, true)
		r[5] = v.ref(subject.OptionalRelation, subject.OptionalRelation != "")
	}
	if i == v.n-1 {
		v.finish()
	}
}
func (v *viewItems) finish() {
	v.cblob = C.malloc(C.size_t(len(v.blob) + 1))
	if len(v.blob) > 0 {
		copy(unsafe.Slice((*byte)(v.cblob), len(v.blob)), v.blob)
	}
	v.items = (*C.acl_check_item_v_t)(C.calloc(C.size_t(v.n), C.size_t(unsafe.Sizeof(C.acl_check_item_v_t{}))))
	out := unsafe.Slice(v.items, v.n)
	for i := range out {
		fields := (*[6]C.acl_str_t)(unsafe.Pointer(&out[i])) // six consecutive {p, n} views (aclgpu.h)
		for f, rf := range v.refs[i] {
			if rf[0] >= 0 {
				fields[f].p = (*C.char)(unsafe.Add(v.cblob, rf[0]))
				fields[f].n = C.size_t(rf[1])
			}
		}
	}
}
func (v *viewItems) free() {
	C.free(unsafe.Pointer(v.items))
	C.free(v.cblob)
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
