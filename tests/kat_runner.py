"""Runs the golden vectors of tests/golden/kats.json against anything that
looks like the permissions client (the oracle binding or the engine client).

Adapter protocol (duck-typed):
  write(updates, preconditions) -> revision          raises err with .code
  check(rt, rid, perm, st, sid, srel) -> (permissionship, err_code)
  lookup(rt, perm, st, sid, srel) -> set[str]
  read(**filter) -> [(rt, rid, rel, st, sid, srel, expires)]
  delete_by_filter(**filter) -> count
  set_now(t)
Optional: check_bulk([(rt, rid, perm, st, sid, srel)]) -> ([perm], [err])
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
PERM = {1: "NO", 2: "HAS", 3: "COND"}
OPS = {"CREATE": 1, "TOUCH": 2, "DELETE": 3}
PRE = {"MUST_NOT_MATCH": 1, "MUST_MATCH": 2}

_REL = re.compile(r"^(?P<rt>.*?):(?P<rid>.*?)#(?P<rel>.*?)@(?P<st>.*?):(?P<sid>.*?)(#(?P<srel>.*?))?$")


def parse_rel(text):
    m = _REL.match(text)
    assert m, text
    return (m["rt"], m["rid"], m["rel"], m["st"], m["sid"], m["srel"] or "")


def parse_subject(text):
    m = re.match(r"^(.*?):(.*?)(#(.*))?$", text)
    return m.group(1), m.group(2), m.group(4) or ""


def fmt_rel(t):
    rt, rid, rel, st, sid, srel = t[:6]
    return f"{rt}:{rid}#{rel}@{st}:{sid}" + (f"#{srel}" if srel else "")


def load_bootstrap():
    with open(os.path.join(HERE, "golden", "bootstrap.json")) as f:
        return json.load(f)


def load_kats():
    with open(os.path.join(HERE, "golden", "kats.json")) as f:
        return json.load(f)["kats"]


def kat_schema(kat):
    if kat.get("schema") == "bootstrap":
        b = load_bootstrap()
        return b["schema"], b["relationships"]
    return kat["schema_text"], []


def _outcome(perm, err):
    return f"ERR:{err}" if err else PERM.get(perm, f"?{perm}")


def _match(expected, got):
    if expected == "DENY":
        return got != "HAS"
    if expected.startswith("ERR:") and expected.endswith("*"):
        return got.startswith("ERR:")
    return expected == got


def run_kat(kat, client):
    """client must already hold the kat's schema + seed relationships."""
    for i, step in enumerate(kat["steps"]):
        where = f"{kat['name']} step {i} {step[0]}"
        kind = step[0]
        if kind == "write":
            ups = [(OPS[u[0]], parse_rel(u[1]), u[2] if len(u) > 2 else 0) for u in step[1]]
            pre = [(PRE[p[0]], p[1]) for p in (step[2] if len(step) > 2 else [])]
            expect = step[3] if len(step) > 3 else None
            try:
                client.write(ups, pre)
                got = "OK"
            except Exception as e:  # noqa: BLE001 - adapters raise their own error type with .code
                code = getattr(e, "code", None)
                assert code is not None, f"{where}: unexpected exception {e!r}"
                got = f"ERR:{code}"
            assert got == (expect or "OK"), f"{where}: expected {expect or 'OK'}, got {got}"
        elif kind == "check":
            perm, err = client.check(*parse_rel(step[1]))
            assert _match(step[2], _outcome(perm, err)), f"{where}: {step[1]} expected {step[2]} got {_outcome(perm, err)}"
        elif kind == "check_raw":
            perm, err = client.check(*step[1])
            assert _match(step[2], _outcome(perm, err)), f"{where}: expected {step[2]} got {_outcome(perm, err)}"
        elif kind == "bulk":
            items = [parse_rel(t) for t in step[1]]
            if hasattr(client, "check_bulk"):
                perms, errs = client.check_bulk(items)
                got = [_outcome(p, e) for p, e in zip(perms, errs)]
            else:
                got = [_outcome(*client.check(*it)) for it in items]
            assert all(_match(e, g) for e, g in zip(step[2], got)) and len(got) == len(step[2]), f"{where}: expected {step[2]} got {got}"
        elif kind == "bulk_err":  # a request the API's validation refuses fails AS A WHOLE (no pairs)
            items = [parse_rel(t) for t in step[1]]
            msg = None
            if hasattr(client, "check_bulk"):
                try:
                    client.check_bulk(items)
                    got = None
                except Exception as e:  # noqa: BLE001
                    got = getattr(e, "code", None)
                    msg = str(e)
            else:  # one-at-a-time adapters: the call fails iff some item is ill-formed
                got = 3 if any(client.check(*it)[1] == 3 for it in items) else None
            assert got == step[2], f"{where}: expected the call to fail with {step[2]}, got {got}"
            # what the ENGINE says about it (step[3], optional): the reference denies everything a failed CheckBulkPermissions asked (check.go:48-52), so the
            # message is all an operator has -- it names the item, the field, the value and the byte that breaks the pattern (the oracles carry no messages)
            if len(step) > 3 and msg is not None and getattr(client, "names_invalid_fields", False):
                assert all(part in msg for part in step[3]), f"{where}: the message {msg!r} should name {step[3]}"
        elif kind == "lookup":
            st, sid, srel = parse_subject(step[3])
            got = client.lookup(step[1], step[2], st, sid, srel)
            assert set(got) == set(step[4]), f"{where}: expected {sorted(step[4])} got {sorted(got)}"
        elif kind == "read":
            got = sorted(fmt_rel(t) for t in client.read(**step[1]))
            assert got == sorted(step[2]), f"{where}: expected {sorted(step[2])} got {got}"
        elif kind == "delete_filter":
            n = client.delete_by_filter(**step[1])
            assert n == step[2], f"{where}: expected {step[2]} deleted, got {n}"
        elif kind == "set_now":
            client.set_now(step[1])
        else:
            raise AssertionError(f"unknown KAT step {kind}")
