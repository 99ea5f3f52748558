"""PreFilter consumers on bytes: acl_prefilter_response against a Python restatement of the reference's filterList /
filterTable / filterObject (pkg/authz/responsefilterer.go:349-416) whose allowed set is built the reference's way -- every id of the
LookupResources result mapped to a NamespacedName by the rule's fromObjectID expressions (pkg/authz/lookups.go:98-131; deploy/rules.yaml:51,
105-106) -- and asked IsAllowed(namespace, name) (lookups.go:25-36).  Needs no GPU: the bitmap is the test's own."""
import json
import random

import numpy as np
import pytest

SCHEMA = """
definition user {}
definition namespace { relation viewer: user
  permission view = viewer }
definition pod { relation viewer: user
  permission view = viewer }
"""


def allowed_set(ids, split):
    """lookups.go:98-131: NameFromObjectID / NamespaceFromObjectID over every HAS_PERMISSION resource id."""
    out = set()
    for rid in ids:
        if split:  # rules.yaml:105-106: split_namespace(resourceId) / split_name(resourceId)
            ns, _, name = rid.partition("/") if "/" in rid else ("", "", rid)
        else:      # rules.yaml:51: {{resourceId}}, no namespace expression
            ns, name = "", rid
        out.add((ns, name))
    return out


def ns_name(meta):
    md = meta.get("metadata") if isinstance(meta, dict) else None
    md = md if isinstance(md, dict) else {}
    g = lambda k: md[k] if isinstance(md.get(k), str) else ""  # noqa: E731
    return g("namespace"), g("name")


def reference_filter(body, allowed, kind):
    """responsefilterer.go:349-416 restated over decoded JSON -> the filtered JSON value."""
    doc = json.loads(body)
    if kind == "list":      # filterList: allowedItems := make([]runtime.Object, 0) ... meta.SetList
        doc["items"] = [it for it in doc["items"] if ns_name(it) in allowed]
    elif kind == "table":   # filterTable: allowedRows := make([]metav1.TableRow, 0)
        doc["rows"] = [r for r in doc["rows"] if ns_name(r["object"]) in allowed]
    return doc


@pytest.fixture()
def eng(aclgpu_lib, monkeypatch):
    import aclgpu
    # the scanners' decoding tests name objects with bytes no API request could carry (multi-byte, escapes): acl_intern takes them raw here
    monkeypatch.setenv("ACL_RAW_INTERN", "1")  # (read at acl_open)
    e = aclgpu.Engine(SCHEMA, store_only=True)
    yield e
    e.close()


def bitmap_of(e, rtype, ids):
    bm = np.zeros((e.object_count(rtype) + 31) // 32 + 1, dtype=np.uint32)
    for rid in ids:
        i = e.find(rtype, rid)
        bm[i >> 5] |= np.uint32(1 << (i & 31))
    return bm


def test_lists_tables_and_objects_match_the_reference_consumers(eng):
    rng = random.Random(7)
    pods = [f"ns{n}/pod-{i}" for n in range(4) for i in range(40)] + ["solo", "ns0/ünï-é", 'ns1/quo"te']
    for p in pods:
        eng.intern("pod", p)
    for trial in range(30):
        allowed_ids = rng.sample(pods, rng.randrange(0, len(pods)))
        bm = bitmap_of(eng, "pod", allowed_ids)
        allowed = allowed_set(allowed_ids, split=True)
        names = rng.sample(pods, rng.randrange(0, 60)) + [f"ns9/never-interned-{trial}", "ns0/pod-1/extra"]
        rng.shuffle(names)
        items = []
        for rid in names:
            ns, _, name = rid.partition("/") if "/" in rid else ("", "", rid)
            md = {"name": name, "labels": {"a": "b"}, "resourceVersion": str(rng.randrange(1 << 30))}
            if ns:
                md["namespace"] = ns
            items.append({"kind": "Pod", "spec": {"containers": [{"name": "c", "args": ["{", "]", "\\"]}]}, "metadata": md})
        if trial % 5 == 0:
            items.append({"metadata": {"name": 7}})          # a name that is not a string: ("", "")
            items.append({"spec": {}})                        # no metadata at all
        ensure_ascii = trial % 2 == 0
        list_body = json.dumps({"kind": "PodList", "apiVersion": "v1", "metadata": {"resourceVersion": "1"}, "items": items}, ensure_ascii=ensure_ascii,
                               indent=None if trial % 3 else 2).encode()
        out, kept, total = eng.prefilter_response("pod", bm, "{{namespacedName}}", eng.BODY_LIST, list_body)
        want = reference_filter(list_body, allowed, "list")
        assert json.loads(out) == want and kept == len(want["items"]) and total == len(items)
        if not want["items"]:
            assert b'"items":[]' in out.replace(b" ", b"").replace(b"\n", b"")
        rows = [{"cells": [it["metadata"].get("name"), "1/1", "Running"], "object": {"kind": "PartialObjectMetadata", "apiVersion": "meta.k8s.io/v1", "metadata": it["metadata"]}}
                for it in items if "metadata" in it]
        table_body = json.dumps({"kind": "Table", "apiVersion": "meta.k8s.io/v1", "columnDefinitions": [{"name": "Name", "type": "string"}], "rows": rows},
                                ensure_ascii=ensure_ascii).encode()
        out, kept, total = eng.prefilter_response("pod", bm, "{{namespacedName}}", eng.BODY_TABLE, table_body)
        want = reference_filter(table_body, allowed, "table")
        assert json.loads(out) == want and kept == len(want["rows"]) and total == len(rows)
        # filterObject: the body unchanged, or "unauthorized"
        for it in items[:6]:
            body = json.dumps(it).encode()
            if ns_name(it) in allowed:
                assert eng.prefilter_response("pod", bm, "{{namespacedName}}", eng.BODY_OBJECT, body) == (body, 1, 1)
            else:
                with pytest.raises(Exception) as x:
                    eng.prefilter_response("pod", bm, "{{namespacedName}}", eng.BODY_OBJECT, body)
                assert x.value.code == 7 and "unauthorized" in str(x.value)


def test_cluster_scoped_ids_and_edge_bodies(eng):
    nss = ["default", "kube-system", "team-a", "team-b"]
    for n in nss:
        eng.intern("namespace", n)
    bm = bitmap_of(eng, "namespace", ["default", "team-b"])
    allowed = allowed_set(["default", "team-b"], split=False)
    body = json.dumps({"kind": "NamespaceList", "items": [{"metadata": {"name": n}} for n in nss + ["ghost"]]}).encode()
    out, kept, total = eng.prefilter_response("namespace", bm, "{{name}}", eng.BODY_LIST, body)
    assert json.loads(out) == reference_filter(body, allowed, "list") and (kept, total) == (2, 5)
    # a namespaced item never matches a cluster-scoped id mapping: IsAllowed("x", "default") is false
    body = json.dumps({"items": [{"metadata": {"name": "default", "namespace": "x"}}]}).encode()
    out, kept, _ = eng.prefilter_response("namespace", bm, "{{namespacedName}}", eng.BODY_LIST, body)
    assert json.loads(out) == {"items": []} and kept == 0
    # nothing to cut: no array, an empty one, the array under another key
    for b in (b'{"kind":"PodList"}', b'{"items":[]}', b'{"rows":[{"object":{"metadata":{"name":"default"}}}]}'):
        assert eng.prefilter_response("namespace", bm, "{{name}}", eng.BODY_LIST, b) == (b, 0, 0)
    # a short bitmap (objects interned after the lookup): bits beyond it are not allowed
    out, kept, _ = eng.prefilter_response("namespace", bm[:0], "{{name}}", eng.BODY_LIST, json.dumps({"items": [{"metadata": {"name": "default"}}]}).encode())
    assert kept == 0
    # errors: the reference's decode failures
    bad = [(eng.BODY_TABLE, b'{"rows":[{"cells":[]}]}'), (eng.BODY_TABLE, b'{"rows":[{"object":null}]}'), (eng.BODY_TABLE, b'{"rows":[{"object":"x"}]}'),
           (eng.BODY_TABLE, b'{"rows":[7]}'), (eng.BODY_LIST, b'{"items":[1]}'), (eng.BODY_LIST, b'{"items":[{}],}'), (eng.BODY_LIST, b'[]'), (eng.BODY_OBJECT, b'{"metadata":'),
           (eng.BODY_LIST, b'{"items":[{"metadata":{"name":"a"}}]} x')]
    for kind, b in bad:
        with pytest.raises(Exception) as x:
            eng.prefilter_response("namespace", bm, "{{name}}", kind, b)
        assert x.value.code == 3, b
    with pytest.raises(Exception) as x:  # a template the engine does not render
        eng.prefilter_response("namespace", bm, "{{metadata.labels.x}}", eng.BODY_OBJECT, b'{"metadata":{"name":"default"}}')
    assert x.value.code == 7


def test_string_scanning_fast_path_decodes_what_json_decodes(eng):
    """The scanner walks string content eight bytes at a time until a quote, a backslash or a control character shows up (engine_list.cpp
    Scanner::string): names with those bytes at every offset of a window, escapes after long plain runs, multi-byte UTF-8 and \\u escapes must
    decode to exactly what encoding/json (here: Python's json) decodes -- the item is found under that name or it is not."""
    rng = random.Random(3)
    alphabet = ['a', 'B', '7', '-', '.', '"', '\\', '/', '\t', '\n', 'é', 'ß', '漢', '😀', ' ', '\x7f']
    names = set()
    for length in list(range(0, 20)) + [31, 32, 33, 63, 64, 65, 200]:
        for _ in range(12):
            names.add("".join(rng.choice(alphabet) for _ in range(length)))
    for pos in range(0, 18):  # one special byte at every offset of a run of plain bytes
        for ch in ('"', '\\', '\n', '😀'):
            names.add("x" * pos + ch + "y" * 9)
    names.discard("")
    names = sorted(names)
    for nm in names:
        eng.intern("pod", nm)
    allowed = set(rng.sample(names, len(names) // 2))
    bm = bitmap_of(eng, "pod", allowed)
    for ensure_ascii in (True, False):
        items = [{"metadata": {"name": nm}, "pad": nm * 2} for nm in names] + [{"metadata": {"name": nm + "?"}} for nm in names[:50]]
        body = json.dumps({"items": items}, ensure_ascii=ensure_ascii).encode()
        out, kept, total = eng.prefilter_response("pod", bm, "{{name}}", eng.BODY_LIST, body)
        assert [it["metadata"]["name"] for it in json.loads(out)["items"]] == [nm for nm in names if nm in allowed] and total == len(items)
    for bad in (b'{"items":[{"metadata":{"name":"abcdefgh\x01ijkl"}}]}', b'{"items":[{"metadata":{"name":"abcdefghijklmnop\\q"}}]}', b'{"items":[{"metadata":{"name":"abcdefghijkl'):
        with pytest.raises(Exception) as x:  # a raw control character, an unknown escape, an unterminated string: invalid JSON
            eng.prefilter_response("pod", bm, "{{name}}", eng.BODY_LIST, bad)
        assert x.value.code == 3


def test_big_lists_and_tables_take_the_parallel_scan_and_match_the_reference(eng):
    """Bodies beyond 512 KB: the element spans come from the parallel index (csrc/json_index.hpp), every element is scanned, tested and spliced by the pool's
    threads -- lists and TABLES (scan_row), with strings that hold quotes, backslashes and brackets, against the same restated consumers; the bitmap test of
    a long name list (acl_bitmap_test_names) against the set."""
    rng = random.Random(11)
    pods = [f"ns{n}/pod-{i}" for n in range(8) for i in range(500)]
    for p in pods:
        eng.intern("pod", p)
    for trial in range(3):
        allowed_ids = rng.sample(pods, rng.randrange(1, len(pods)))
        bm = bitmap_of(eng, "pod", allowed_ids)
        allowed = allowed_set(allowed_ids, split=True)
        names = [rng.choice(pods) for _ in range(3500)] + [f"ns9/never-{trial}"]
        items = []
        for k, rid in enumerate(names):
            ns, _, name = rid.partition("/")
            items.append({"kind": "Pod", "metadata": {"name": name, "namespace": ns, "annotations": {"cfg": json.dumps({"a": ["}", "]", "\\", k]}), "pad": "z" * (300 + k % 64)}},
                          "spec": {"containers": [{"name": "c", "args": ["{", "]", "\\\"", ","]}]}})
        list_body = json.dumps({"kind": "PodList", "apiVersion": "v1", "metadata": {"resourceVersion": "1"}, "items": items}, indent=None if trial else 1).encode()
        assert len(list_body) > (1 << 20)
        out, kept, total = eng.prefilter_response("pod", bm, "{{namespacedName}}", eng.BODY_LIST, list_body)
        want = reference_filter(list_body, allowed, "list")
        assert json.loads(out) == want and kept == len(want["items"]) and total == len(items)
        rows = [{"cells": [it["metadata"]["name"], "1/1"], "object": {"kind": "PartialObjectMetadata", "metadata": it["metadata"]}} for it in items]
        table_body = json.dumps({"kind": "Table", "columnDefinitions": [{"name": "Name"}], "rows": rows}).encode()
        out, kept, total = eng.prefilter_response("pod", bm, "{{namespacedName}}", eng.BODY_TABLE, table_body)
        want = reference_filter(table_body, allowed, "table")
        assert json.loads(out) == want and kept == len(want["rows"]) and total == len(rows)
        got = eng.bitmap_test_names("pod", bm, names)
        assert got.tolist() == [tuple(n.split("/", 1)) in allowed for n in names]
        # one broken element in the middle of a big body fails the call as it fails a small one
        broken = list_body.replace(b'"kind": "Pod"', b'"kind": Pod', 1) if b'"kind": "Pod"' in list_body else list_body.replace(b'"kind":"Pod"', b'"kind":Pod', 1)
        with pytest.raises(Exception) as x:
            eng.prefilter_response("pod", bm, "{{namespacedName}}", eng.BODY_LIST, broken)
        assert x.value.code == 3


def test_bitmap_names_in_blocks(eng):
    """acl_bitmap_names: the LookupResources stream's names a block per call -- every set bit once, in id order, whatever the block and buffer sizes; ids that
    no object carries come out as empty names (as the per-id call's -1)."""
    rng = random.Random(3)
    pods = [f"ns{n}/pod-{i}" + "x" * rng.randrange(0, 40) for n in range(5) for i in range(300)] + ["z" * 1024]
    for p in pods:
        eng.intern("pod", p)
    allowed = sorted(rng.sample(pods, 700), key=lambda p: eng.find("pod", p))
    bm = bitmap_of(eng, "pod", allowed)
    for block, buf in ((512, 1 << 16), (1, 1024), (7, 1500), (4096, 1 << 20)):
        assert eng.bitmap_names("pod", bm, block, buf) == allowed
    bm2 = np.concatenate([bm, np.array([0, 5], dtype=np.uint32)])  # two bits beyond every id: nameless
    assert eng.bitmap_names("pod", bm2) == allowed + ["", ""]
    assert eng.bitmap_names("pod", np.zeros(4, dtype=np.uint32)) == []
    with pytest.raises(Exception) as x:
        eng.bitmap_names("pod", bm, 16, 512)
    assert x.value.code == 3
