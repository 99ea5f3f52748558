// refcheck -- replays the cases of tests/ref_cases.py through the REFERENCE's own embedded SpiceDB and writes the
// fixtures tests/golden/ref_<case>.json that pin the CPU oracle (and the GPU engine) to the real thing.
//
// TEST INFRASTRUCTURE, UNBUILT in this repository's build environment: there is no Go toolchain there and the
// arithmetic it calls (github.com/authzed/spicedb, go.mod:9 of the reference) is not vendored under /root/reference.
// It is committed so that the pin is one command away wherever Go and the module cache exist:
//
//	python tools/dump_ref_inputs.py            # oracle/_ref/inputs/<case>/{schema.zed,relationships.txt,checks.txt,lookups.txt}
//	make -C oracle ref REF=/path/to/spicedb-kubeapi-proxy   # builds oracle/_ref/refcheck inside the reference module, runs it
//	python -m pytest tests/test_ref_fixtures.py            # oracle (CPU) and engine (GPU) vs the fixtures
//
// What it exercises is exactly the reference's path (SURVEY.md 8(a)): the server is built by spicedb.NewServer
// (pkg/spicedb/spicedb.go:18-71 -- memdb, dispatch depth 50, caches off, expiration on), requests are the ones
// pkg/authz builds: CheckBulkPermissions with full consistency (check.go:23-48), LookupResources without limit or cursor
// (lookups.go:49-65).  Nothing here is copied from the reference; it only calls its exported API.
package main

import (
	"bufio"
	"context"
	"crypto/sha256"
	"encoding/hex"
	"encoding/json"
	"errors"
	"flag"
	"fmt"
	"io"
	"os"
	"path/filepath"
	"regexp"
	"sort"
	"strings"

	v1 "github.com/authzed/authzed-go/proto/authzed/api/v1"
	"google.golang.org/grpc"
	"google.golang.org/grpc/credentials/insecure"
	"google.golang.org/grpc/status"

	"github.com/authzed/spicedb-kubeapi-proxy/pkg/spicedb"
)

// the tuple grammar of pkg/rules/rules.go:1053-1055
var relRe = regexp.MustCompile(`^(.*?):(.*?)#(.*?)@(.*?):(.*?)(#(.*?))?$`)
var lookupRe = regexp.MustCompile(`^(.*?)#(.*?)@(.*?):(.*?)(#(.*?))?$`)

type fixture struct {
	Case          string           `json:"case"`
	Source        string           `json:"source"`
	Relationships int              `json:"relationships"`
	ChecksSHA256  string           `json:"checks_sha256"` // of checks.txt: the consumer regenerates the inputs and compares
	Perm          string           `json:"perm"`          // one digit per check: Permissionship (1 NO, 2 HAS, 3 CONDITIONAL), 0 = the pair carried an error
	ErrCodes      map[string]int32 `json:"err_codes"`     // check index -> gRPC code of the pair's error
	Lookups       [][]string       `json:"lookups"`       // per lookups.txt line: sorted resource ids with HAS_PERMISSION
	// requests.txt (optional; API validation, tests/ref_cases.py `requests`): one outcome per request, in file order --
	// "check" / "write": the call's gRPC code ("0" = OK; a check adds ":<permissionship>"); "bulk": "error:<code>" when the CALL failed,
	// else one digit per item as in `perm` ("0" = the pair carried an error)
	Requests []string `json:"requests,omitempty"`
}

func lines(path string) ([]string, []byte, error) {
	raw, err := os.ReadFile(path)
	if err != nil {
		return nil, nil, err
	}
	var out []string
	sc := bufio.NewScanner(strings.NewReader(string(raw)))
	sc.Buffer(make([]byte, 1<<20), 1<<20)
	for sc.Scan() {
		if l := strings.TrimSpace(sc.Text()); l != "" {
			out = append(out, l)
		}
	}
	return out, raw, sc.Err()
}

func subject(t, id, rel string) *v1.SubjectReference {
	return &v1.SubjectReference{Object: &v1.ObjectReference{ObjectType: t, ObjectId: id}, OptionalRelation: rel}
}

func full() *v1.Consistency {
	return &v1.Consistency{Requirement: &v1.Consistency_FullyConsistent{FullyConsistent: true}}
}

func runCase(ctx context.Context, dir, outDir string) error {
	name := filepath.Base(dir)
	schema, err := os.ReadFile(filepath.Join(dir, "schema.zed"))
	if err != nil {
		return err
	}
	// bootstrap = schema only (the YAML shape of pkg/spicedb/bootstrap.yaml); relationships go through WriteRelationships
	// in chunks of <= 1000 updates (spicedb.go:35) like the proxy's own writes (activity.go:60)
	var yaml strings.Builder
	yaml.WriteString("schema: |-\n")
	for _, l := range strings.Split(string(schema), "\n") {
		yaml.WriteString("  " + l + "\n")
	}
	yaml.WriteString("relationships: |\n")
	cctx, cancel := context.WithCancel(ctx)
	defer cancel()
	srv, err := spicedb.NewServer(cctx, "", map[string][]byte{"bootstrap.yaml": []byte(yaml.String())})
	if err != nil {
		return fmt.Errorf("%s: NewServer: %w", name, err)
	}
	go func() { _ = srv.Run(cctx) }()
	conn, err := srv.NewClient(grpc.WithTransportCredentials(insecure.NewCredentials()))
	if err != nil {
		return fmt.Errorf("%s: NewClient: %w", name, err)
	}
	defer conn.Close()
	client := v1.NewPermissionsServiceClient(conn)

	rels, _, err := lines(filepath.Join(dir, "relationships.txt"))
	if err != nil {
		return err
	}
	for i := 0; i < len(rels); i += 1000 {
		j := min(i+1000, len(rels))
		ups := make([]*v1.RelationshipUpdate, 0, j-i)
		for _, l := range rels[i:j] {
			m := relRe.FindStringSubmatch(l)
			if m == nil {
				return fmt.Errorf("%s: bad relationship %q", name, l)
			}
			ups = append(ups, &v1.RelationshipUpdate{Operation: v1.RelationshipUpdate_OPERATION_TOUCH, Relationship: &v1.Relationship{
				Resource: &v1.ObjectReference{ObjectType: m[1], ObjectId: m[2]}, Relation: m[3], Subject: subject(m[4], m[5], m[7])}})
		}
		if _, err := client.WriteRelationships(ctx, &v1.WriteRelationshipsRequest{Updates: ups}); err != nil {
			return fmt.Errorf("%s: WriteRelationships[%d:%d]: %w", name, i, j, err)
		}
	}

	checks, rawChecks, err := lines(filepath.Join(dir, "checks.txt"))
	if err != nil {
		return err
	}
	sum := sha256.Sum256(rawChecks)
	fx := fixture{Case: name, Source: "embedded SpiceDB via pkg/spicedb/spicedb.go:18-71; CheckBulkPermissions check.go:48; LookupResources lookups.go:65",
		Relationships: len(rels), ChecksSHA256: hex.EncodeToString(sum[:]), ErrCodes: map[string]int32{}}
	perm := make([]byte, len(checks))
	for i := 0; i < len(checks); i += 500 {
		j := min(i+500, len(checks))
		items := make([]*v1.CheckBulkPermissionsRequestItem, 0, j-i)
		for _, l := range checks[i:j] {
			m := relRe.FindStringSubmatch(l)
			if m == nil {
				return fmt.Errorf("%s: bad check %q", name, l)
			}
			items = append(items, &v1.CheckBulkPermissionsRequestItem{Resource: &v1.ObjectReference{ObjectType: m[1], ObjectId: m[2]}, Permission: m[3],
				Subject: subject(m[4], m[5], m[7])})
		}
		resp, err := client.CheckBulkPermissions(ctx, &v1.CheckBulkPermissionsRequest{Consistency: full(), Items: items})
		if err != nil {
			return fmt.Errorf("%s: CheckBulkPermissions[%d:%d]: %w", name, i, j, err)
		}
		if len(resp.Pairs) != j-i {
			return fmt.Errorf("%s: %d pairs for %d items", name, len(resp.Pairs), j-i)
		}
		for k, p := range resp.Pairs { // pair k answers item k: check.go:54-57
			if e := p.GetError(); e != nil {
				perm[i+k] = '0'
				fx.ErrCodes[fmt.Sprint(i+k)] = e.Code
			} else {
				perm[i+k] = byte('0' + int(p.GetItem().Permissionship))
			}
		}
	}
	fx.Perm = string(perm)

	lks, _, err := lines(filepath.Join(dir, "lookups.txt"))
	if err != nil {
		return err
	}
	for _, l := range lks {
		m := lookupRe.FindStringSubmatch(l)
		if m == nil {
			return fmt.Errorf("%s: bad lookup %q", name, l)
		}
		stream, err := client.LookupResources(ctx, &v1.LookupResourcesRequest{Consistency: full(), ResourceObjectType: m[1], Permission: m[2],
			Subject: subject(m[3], m[4], m[6])})
		if err != nil {
			return fmt.Errorf("%s: LookupResources(%s): %w", name, l, err)
		}
		ids := []string{}
		for {
			r, err := stream.Recv()
			if errors.Is(err, io.EOF) {
				break
			}
			if err != nil {
				// a lookup that ends in an error (e.g. depth) pins an empty/partial set: record the code next to it
				ids = append(ids, "!error:"+status.Code(err).String())
				break
			}
			if r.Permissionship == v1.LookupPermissionship_LOOKUP_PERMISSIONSHIP_HAS_PERMISSION { // lookups.go:85-88
				ids = append(ids, r.ResourceObjectId)
			}
		}
		sort.Strings(ids)
		fx.Lookups = append(fx.Lookups, ids)
	}
	// requests.txt: `check <tuple>` | `write <tuple>` (one TOUCH) | `bulk <tuple> <tuple> ...` -- whole requests whose ERROR BEHAVIOUR is the
	// thing pinned (ill-formed ids, `*`, unknown names: InvalidArgument of the call vs an error inside a pair).  Replayed last: the
	// writes that succeed change the store.
	if reqs, _, err := lines(filepath.Join(dir, "requests.txt")); err == nil {
		item := func(l string) (*v1.CheckBulkPermissionsRequestItem, error) {
			m := relRe.FindStringSubmatch(l)
			if m == nil {
				return nil, fmt.Errorf("%s: bad request tuple %q", name, l)
			}
			return &v1.CheckBulkPermissionsRequestItem{Resource: &v1.ObjectReference{ObjectType: m[1], ObjectId: m[2]}, Permission: m[3], Subject: subject(m[4], m[5], m[7])}, nil
		}
		for _, l := range reqs {
			f := strings.Fields(l)
			if len(f) < 2 {
				return fmt.Errorf("%s: bad request line %q", name, l)
			}
			switch f[0] {
			case "check":
				it, err := item(f[1])
				if err != nil {
					return err
				}
				r, err := client.CheckPermission(ctx, &v1.CheckPermissionRequest{Consistency: full(), Resource: it.Resource, Permission: it.Permission, Subject: it.Subject})
				if err != nil {
					fx.Requests = append(fx.Requests, fmt.Sprint(int(status.Code(err))))
				} else {
					fx.Requests = append(fx.Requests, fmt.Sprintf("0:%d", int(r.Permissionship)))
				}
			case "write":
				it, err := item(f[1])
				if err != nil {
					return err
				}
				_, err = client.WriteRelationships(ctx, &v1.WriteRelationshipsRequest{Updates: []*v1.RelationshipUpdate{{Operation: v1.RelationshipUpdate_OPERATION_TOUCH,
					Relationship: &v1.Relationship{Resource: it.Resource, Relation: it.Permission, Subject: it.Subject}}}})
				fx.Requests = append(fx.Requests, fmt.Sprint(int(status.Code(err))))
			case "bulk":
				var items []*v1.CheckBulkPermissionsRequestItem
				for _, t := range f[1:] {
					it, err := item(t)
					if err != nil {
						return err
					}
					items = append(items, it)
				}
				resp, err := client.CheckBulkPermissions(ctx, &v1.CheckBulkPermissionsRequest{Consistency: full(), Items: items})
				if err != nil {
					fx.Requests = append(fx.Requests, fmt.Sprintf("error:%d", int(status.Code(err))))
					break
				}
				digits := make([]byte, len(resp.Pairs))
				for k, p := range resp.Pairs {
					if p.GetError() != nil {
						digits[k] = '0'
					} else {
						digits[k] = byte('0' + int(p.GetItem().Permissionship))
					}
				}
				fx.Requests = append(fx.Requests, string(digits))
			default:
				return fmt.Errorf("%s: unknown request kind %q", name, f[0])
			}
		}
	}
	out, err := json.Marshal(fx)
	if err != nil {
		return err
	}
	return os.WriteFile(filepath.Join(outDir, "ref_"+name+".json"), append(out, '\n'), 0o644)
}

func main() {
	in := flag.String("inputs", "oracle/_ref/inputs", "directory written by tools/dump_ref_inputs.py")
	out := flag.String("out", "tests/golden", "where ref_<case>.json go")
	flag.Parse()
	dirs, err := filepath.Glob(filepath.Join(*in, "*"))
	if err != nil || len(dirs) == 0 {
		fmt.Fprintln(os.Stderr, "no cases under", *in)
		os.Exit(2)
	}
	for _, d := range dirs {
		if err := runCase(context.Background(), d, *out); err != nil {
			fmt.Fprintln(os.Stderr, "FAILED:", err)
			os.Exit(1)
		}
		fmt.Println("wrote", filepath.Join(*out, "ref_"+filepath.Base(d)+".json"))
	}
}
