"""Relationship text helpers: the grammar of reference pkg/rules/rules.go:1053-1055
(`ParseRelSring`), which is also what tupleSet rule outputs and bootstrap files use."""
from __future__ import annotations

import re

_REL = re.compile(r"^(?P<rt>.*?):(?P<rid>.*?)#(?P<rel>.*?)@(?P<st>.*?):(?P<sid>.*?)(#(?P<srel>.*?))?$")


def parse_relationship(text: str):
    """'type:id#rel@type:id[#rel]' -> (rtype, rid, rel, stype, sid, srel) with srel '' when absent."""
    m = _REL.match(text)
    if not m:
        raise ValueError(f"invalid relationship text {text!r}")
    return (m["rt"], m["rid"], m["rel"], m["st"], m["sid"], m["srel"] or "")


def format_relationship(t) -> str:
    rt, rid, rel, st, sid, srel = t[:6]
    return f"{rt}:{rid}#{rel}@{st}:{sid}" + (f"#{srel}" if srel else "")
