"""List-level PostFilter: acl_filter_list_response, the reference's filterListResponse
(pkg/authz/postfilter.go:17-55), fed with the reference's own test vectors (pkg/authz/postfilter_test.go:181-317) and checked
against a line-by-line Python restatement of the reference algorithm (json decode -> per-item checks -> json encode) whose
checks are answered by the CPU oracle."""
import json
import random

import pytest

from oracle import orc

SCHEMA = """
definition user {}
definition namespace { relation viewer: user
  permission view = viewer }
definition pod {
  relation namespace: namespace
  relation viewer: user
  relation creator: user
  permission view = viewer + creator + namespace->view
  permission edit = creator
}
"""


def reference_filter_list_response(body: bytes, templates, user, check):
    """postfilter.go:17-182 restated: returns the filtered body as a JSON VALUE (None = body unchanged)."""
    doc = json.loads(body)
    items = doc.get("items") if isinstance(doc, dict) else None
    if not isinstance(items, list) or not items:
        return None
    bulk, item_reqs = [], {}
    for i, item in enumerate(items):
        if not isinstance(item, dict):
            continue
        name = ns = ""
        md = item.get("metadata")
        if isinstance(md, dict):
            if isinstance(md.get("name"), str):
                name = md["name"]
            if isinstance(md.get("namespace"), str):
                ns = md["namespace"]
        for t in templates:
            text = t.replace("{{name}}", name).replace("{{namespace}}", ns).replace("{{namespacedName}}", f"{ns}/{name}" if ns else name).replace("{{user.name}}", user)
            if "{{" in text:
                continue  # failed to resolve: no check (postfilter.go:92-95)
            try:
                rel = orc.parse_rel(text)
            except ValueError:
                continue
            item_reqs.setdefault(i, []).append(len(bulk))
            bulk.append(rel)
    if not bulk:
        return None
    answers = [check(*r) for r in bulk]
    if any(a[1] == 3 for a in answers):
        # one ill-formed item (an empty or non-conforming object id: validate.hpp) fails the CheckBulkPermissions CALL, and the reference
        # returns that error instead of a filtered list (postfilter.go:134-137)
        return "INVALID_ARGUMENT"
    allowed = [it for i, it in enumerate(items) if i not in item_reqs or all(answers[k] == (2, 0) for k in item_reqs[i])]
    doc["items"] = allowed if allowed else None  # appending to a nil slice: nothing allowed marshals as null
    return doc


def pods(names, ns="default"):
    return {"apiVersion": "v1", "kind": "PodList", "items": [{"metadata": {"name": n, "namespace": ns}} for n in names]}


@pytest.mark.gpu
def test_reference_vectors(aclgpu_lib):
    """TestFilterListResponse (postfilter_test.go:181-243) and TestFilterItemsWithBulkPermissions (:248-317): the mock client
    answers HAS for pod1 / testpod1 and NO for pod2 / testpod2; here the relationships say the same."""
    import aclgpu
    with aclgpu.Engine(SCHEMA) as e:
        e.touch(("pod", "pod1", "viewer", "user", "testuser", ""), ("pod", "testpod1", "creator", "user", "testuser", ""))
        tpl = ["pod:{{name}}#view@user:{{user.name}}"]
        out, kept, total = e.filter_list_response(json.dumps(pods(["pod1", "pod2"])).encode(), tpl, "testuser")
        doc = json.loads(out)
        assert (kept, total) == (1, 2) and [i["metadata"]["name"] for i in doc["items"]] == ["pod1"]
        assert doc["apiVersion"] == "v1" and doc["kind"] == "PodList"
        out, kept, total = e.filter_list_response(json.dumps(pods(["testpod1", "testpod2"])).encode(), tpl, "testuser")
        assert [i["metadata"]["name"] for i in json.loads(out)["items"]] == ["testpod1"]
        # empty items / no items array: the body comes back untouched (postfilter.go:26-35)
        for body in (b'{"kind":"PodList","items":[]}', b'{"kind":"Status","code":404}', b'{"items":"nope"}'):
            out, _k, _t = e.filter_list_response(body, tpl, "testuser")
            assert out == body
        # nothing allowed: the reference's nil slice marshals as null
        out, kept, _t = e.filter_list_response(json.dumps(pods(["pod2"])).encode(), tpl, "testuser")
        assert kept == 0 and json.loads(out)["items"] is None
        with pytest.raises(aclgpu.AclError) as ei:
            e.filter_list_response(b'{"items": [', tpl, "testuser")
        assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT
        # rules.NewResolveInput's fallbacks (rules.go:315-342): an item WITHOUT metadata.namespace in a namespaced list takes the request's namespace;
        # without the request it resolves to another id and is dropped
        e.touch(("pod", "team-a/web", "creator", "user", "testuser", ""), ("namespace", "team-a", "viewer", "user", "testuser", ""))
        tpl2 = ["pod:{{namespacedName}}#view@user:{{user.name}}"]
        body = json.dumps({"kind": "PodList", "items": [{"metadata": {"name": "web"}}, {"metadata": {"name": "web", "namespace": "team-b"}}]}).encode()
        out, kept, total = e.filter_list_response(body, tpl2, "testuser", request=("", "team-a", "pods"))
        assert (kept, total) == (1, 2) and json.loads(out)["items"] == [{"metadata": {"name": "web"}}]
        out, kept, _t = e.filter_list_response(body, tpl2, "testuser")
        assert kept == 0
        # ... and a request on the `namespaces` resource carries the namespace name in BOTH fields: the namespace is cleared (cluster scoped)
        nsl = json.dumps({"kind": "NamespaceList", "items": [{"metadata": {"name": "team-a"}}, {"metadata": {"name": "team-z"}}]}).encode()
        out, kept, _t = e.filter_list_response(nsl, ["namespace:{{namespacedName}}#view@user:{{user.name}}"], "testuser", request=("team-a", "team-a", "namespaces"))
        assert kept == 1 and [i["metadata"]["name"] for i in json.loads(out)["items"]] == ["team-a"]


@pytest.mark.gpu
def test_matches_reference_algorithm_on_random_lists(aclgpu_lib):
    import aclgpu
    rng = random.Random(0x5ACE)
    o = orc.Oracle(SCHEMA)
    with aclgpu.Engine(SCHEMA) as e:
        rels = []
        for n in range(8):
            rels.append(("namespace", f"ns{n}", "viewer", "user", f"u{n % 3}", ""))
        for p in range(300):
            ns = f"ns{rng.randrange(8)}"
            rels.append(("pod", f"{ns}/p{p}", "namespace", "namespace", ns, ""))
            rels.append(("pod", f"{ns}/p{p}", "creator", "user", f"u{rng.randrange(6)}", ""))
            if rng.random() < 0.3:
                rels.append(("pod", f"{ns}/p{p}", "viewer", "user", f"u{rng.randrange(6)}", ""))
        for i in range(0, len(rels), 500):
            o.write([(orc.OP_TOUCH, r) for r in rels[i:i + 500]])
            e.write([(aclgpu.OP_TOUCH, r) for r in rels[i:i + 500]])
        pod_ids = [r[1] for r in rels if r[0] == "pod" and r[2] == "namespace"]
        weird = [7, "str", None, {"no": "metadata"}, {"metadata": "not-an-object"}, {"metadata": {"name": 5}}, {"metadata": {"name": "qé\"x\\y", "namespace": "ns1"}},
                 {"metadata": {"name": "dup", "namespace": "ns0"}, "spec": {"containers": [{"image": "a:b", "ports": [1, 2.5, -3e2]}], "x": [[], {}, [[]]]}},
                 {"metadata": {"name": "a:b", "namespace": "ns1"}}, {"metadata": {"name": "x#view@user:u1", "namespace": "ns2"}}]  # (names that hold the grammar's separators)
        for trial in range(44):
            items = []
            # (the last trials are long lists of padded items: bodies of 1-2 MB, which take the parallel element index and the parallel resolve / splice)
            for k in range(rng.randrange(0, 60) if trial < 40 else 2500):
                if rng.random() < 0.15:
                    items.append(rng.choice(weird))
                else:
                    ns, name = rng.choice(pod_ids).split("/")
                    items.append({"metadata": {"name": name, "namespace": ns, "labels": {"a": "b"}}, "status": {"phase": "Running"}})
                    if trial >= 40:
                        items[-1]["spec"] = {"pad": "x\\\"}],[{" * (k % 7) + "y" * 400}
            doc = {"kind": "PodList", "metadata": {"resourceVersion": str(trial)}, "items": items, "apiVersion": "v1"}
            body = json.dumps(doc, indent=rng.choice([None, 1]), ensure_ascii=rng.random() < 0.5).encode()
            tpls = rng.choice([["pod:{{namespacedName}}#view@user:{{user.name}}"], ["pod:{{namespacedName}}#view@user:{{user.name}}", "pod:{{namespacedName}}#edit@user:{{user.name}}"],
                               ["namespace:{{namespace}}#view@user:{{user.name}}"], ["pod:{{ namespacedName }}#view@user:{{user.name}}", "pod:{{unknownVar}}#view@user:x"], []])
            user = f"u{rng.randrange(6)}"
            want = reference_filter_list_response(body, [t.replace("{{ namespacedName }}", "{{namespacedName}}") for t in tpls], user, o.check)
            if want == "INVALID_ARGUMENT":
                with pytest.raises(aclgpu.AclError) as ei:
                    e.filter_list_response(body, tpls, user)
                assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT, (trial, tpls)
                continue
            out, kept, total = e.filter_list_response(body, tpls, user)
            if want is None:
                assert out == body, (trial, tpls)
            else:
                assert json.loads(out) == want, (trial, tpls, user)
                assert kept == len(want["items"] or []) and total == len(items)


def test_unchanged_and_invalid_bodies_need_no_gpu(aclgpu_lib):
    """The scan half runs anywhere: bodies the reference returns untouched, and bodies json.Unmarshal rejects."""
    import aclgpu
    e = aclgpu.Engine(SCHEMA, store_only=True)
    tpl = ["pod:{{name}}#view@user:{{user.name}}"]
    for body in (b'{"kind":"PodList","items":[]}', b' { "a" : [1, {"b": null}], "items" : 3 } ', b'{"items":[1],"items":{}}'):
        out, _k, _t = e.filter_list_response(body, tpl, "u")
        assert out == body
    for ok_numbers in (b'{"a":[0,-0,1,-12,3.5,1e9,1E+9,2.5e-3,0.0,10.01],"items":[]}',):
        out, _k, _t = e.filter_list_response(ok_numbers, tpl, "u")
        assert out == ok_numbers
    for bad in (b'', b'[1,2]', b'{"items": [}', b'{"a": tru}', b'{"a": "\\q"}', b'{"a":1} trailing', b'{"a":"\x01"}',
                # numbers encoding/json rejects (they used to be spliced through)
                b'{"a":+1}', b'{"a":1.2.3}', b'{"a":--1}', b'{"a":1e}', b'{"a":01}', b'{"a":.5}', b'{"a":1.}', b'{"a":-}', b'{"a":1e+}', b'{"items":[{"x":1.2.3}]}'):
        with pytest.raises(aclgpu.AclError) as ei:
            e.filter_list_response(bad, tpl, "u")
        assert ei.value.code == aclgpu.ERR_INVALID_ARGUMENT
    # a list with items but a store-only engine: the check itself is what needs the device
    with pytest.raises(aclgpu.AclError) as ei:
        e.filter_list_response(b'{"items":[{"metadata":{"name":"p"}}]}', tpl, "u")
    assert ei.value.code == aclgpu.ERR_UNAVAILABLE
    e.close()


def _random_value(rng, depth=0):
    """JSON values whose strings are full of what a naive splitter trips over: quotes, backslash runs of every parity, brackets, commas."""
    k = rng.random()
    if depth > 4 or k < 0.35:
        if rng.random() < 0.6:
            pieces = ['"', "\\", "\\\\", '\\"', "{", "}", "[", "]", ",", ":", " ", "a", "é", "\n", "\t", "\\\\\\", '","', '"},{"', "x" * rng.randrange(0, 70)]
            return "".join(rng.choice(pieces) for _ in range(rng.randrange(0, 12)))
        return rng.choice([0, -1, 1.5e-7, True, False, None, 12345678901234])
    if k < 0.7:
        return {(str(_random_value(rng, 9)) + str(j) if rng.random() < 0.3 else f"k{j}"): _random_value(rng, depth + 1) for j in range(rng.randrange(0, 5))}
    return [_random_value(rng, depth + 1) for _ in range(rng.randrange(0, 5))]


def test_parallel_element_index_matches_a_json_decoder(aclgpu_lib):
    """csrc/json_index.hpp through acl_selfcheck_json_array: the element spans of an `items` array found chunk by chunk (every chunk indexed under both
    in-string hypotheses, escapes carried across 64-byte blocks and across chunks) must be exactly the elements json.loads sees -- chunk sizes from one
    64-byte block up, documents whose strings hold quotes, backslash runs, brackets and commas, different separators and indentation; broken documents fail."""
    import aclgpu
    rng = random.Random(0x5ACE0115)
    e = aclgpu.Engine(SCHEMA, store_only=True)
    try:
        for trial in range(160):
            items = [_random_value(rng) for _ in range(rng.randrange(0, 40))]
            if trial % 7 == 0:
                items = [{"metadata": {"name": f"p{j}", "annotations": {"last-applied": json.dumps({"a": [1, {"b": "}]"}], "c": "\\"})}}, "spec": {"x": [j, [j]]}} for j in range(rng.randrange(1, 60))]
            kw = rng.choice([dict(separators=(",", ":")), dict(), dict(indent=1), dict(separators=(" ,\n", " : ")), dict(ensure_ascii=False, separators=(",", ":"))])
            doc = {"kind": "List", "note": 'tricky "[{" \\', "items": items, "tail": [1, {"z": "]"}]}
            body = json.dumps(doc, **kw).encode()
            arr_open = body.index(b'"items"') + len(b'"items"')
            arr_open = body.index(b"[", arr_open)
            want = [json.dumps(v, **kw).encode() for v in items]
            for chunk in (64, 128, 192, 448, 4096, 0):
                spans, close = e.selfcheck_json_array(body, arr_open, chunk)
                got = [body[b:e_] for b, e_ in spans]
                assert len(got) == len(want), (trial, chunk, len(got), len(want))
                for g, w_ in zip(got, want):
                    assert json.loads(g) == json.loads(w_) and g == g.strip(), (trial, chunk, g[:80])
                assert body[close:close + 1] == b"]" and json.loads(b"[" + b",".join(got) + b"]") == items
            # a document cut short, an element that is not a value, a stray separator: all refused, at every chunk size
            for broken in (body[:max(arr_open + 1, len(body) // 2)], body[:arr_open + 1] + b"1,,2" + body[arr_open + 1:], body[:arr_open + 1] + b'{"a":tru},' + body[arr_open + 1:]):
                for chunk in (64, 4096):
                    with pytest.raises(aclgpu.AclError):
                        e.selfcheck_json_array(broken, arr_open, chunk)
    finally:
        e.close()
