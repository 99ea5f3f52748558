#!/usr/bin/env python3
"""Writes the inputs of oracle/ref_spicedb (the embedded-SpiceDB replay) under oracle/_ref/inputs/<case>/:
schema.zed, relationships.txt, checks.txt (`type:id#perm@type:id[#rel]` per line, the tuple grammar of
pkg/rules/rules.go:1053-1055) and lookups.txt (`type#perm@type:id[#rel]`).  Deterministic: tests/ref_cases.py."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
from tests import ref_cases  # noqa: E402


def fmt(t):
    rt, rid, rel, st, sid, srel = t
    return f"{rt}:{rid}#{rel}@{st}:{sid}" + (f"#{srel}" if srel else "")


def main():
    out = os.path.join(ROOT, "oracle", "_ref", "inputs")
    for c in ref_cases.cases():
        d = os.path.join(out, c["name"])
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "schema.zed"), "w").write(c["schema"])
        open(os.path.join(d, "relationships.txt"), "w").write("".join(fmt(r) + "\n" for r in c["relationships"]))
        checks = "".join(fmt(q) + "\n" for q in c["checks"])
        open(os.path.join(d, "checks.txt"), "w").write(checks)
        open(os.path.join(d, "lookups.txt"), "w").write("".join(f"{rt}#{p}@{st}:{sid}" + (f"#{srel}" if srel else "") + "\n" for rt, p, st, sid, srel in c["lookups"]))
        if c.get("requests"):  # whole requests whose error behaviour is pinned: `check t` | `write t` | `bulk t t ...`
            open(os.path.join(d, "requests.txt"), "w").write("".join(kind + " " + " ".join(fmt(t) for t in ts) + "\n" for kind, ts in c["requests"]))
        print(c["name"], len(c["relationships"]), "relationships,", len(c["checks"]), "checks,", len(c["lookups"]), "lookups, sha256(checks)",
              hashlib.sha256(checks.encode()).hexdigest()[:16])


if __name__ == "__main__":
    main()
