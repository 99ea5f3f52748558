#!/usr/bin/env python3
"""Hard-wraps the prose of a markdown file at a column (tables, fenced code, headings and link-only lines stay as they are; list items keep
their hanging indent).  tools/wrap_md.py FILE [--width 120] rewrites FILE in place; --check exits 1 if a prose line is longer."""
import re
import sys
import textwrap


def wrap(text: str, width: int) -> str:
    out, fence = [], False
    para, indent0, indent = [], "", ""

    def flush():
        nonlocal para
        if para:
            body = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=indent0, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
            para = []

    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        if fence or line.startswith("|") or line.startswith("#") or not line.strip():
            flush()
            out.append(line)
            continue
        m = re.match(r"^(\s*)([-*]|\d+[a-z]?\.)\s+", line)
        if m:  # a new list item
            flush()
            indent0 = m.group(0)
            indent = " " * len(indent0)
            para = [line[len(indent0):]]
            continue
        if para and (line.startswith(indent) or not indent):
            para.append(line)
            continue
        if not para:
            lead = re.match(r"^\s*", line).group(0)
            indent0 = indent = lead
            para = [line]
            continue
        flush()
        lead = re.match(r"^\s*", line).group(0)
        indent0 = indent = lead
        para = [line]
    flush()
    return "\n".join(out)


def main():
    args = sys.argv[1:]
    width = 120
    if "--width" in args:
        width = int(args[args.index("--width") + 1])
    path = [a for a in args if not a.startswith("--") and not a.isdigit()][0]
    text = open(path).read()
    if "--check" in args:
        fence = False
        bad = []
        for i, line in enumerate(text.split("\n"), 1):
            if line.lstrip().startswith("```"):
                fence = not fence
            if not fence and not line.startswith("|") and not line.startswith("#") and len(line) > width and " " in line.strip()[: width]:
                bad.append(i)
        if bad:
            print(f"{path}: {len(bad)} prose lines longer than {width} columns, first at line {bad[0]}")
            sys.exit(1)
        return
    open(path, "w").write(wrap(text, width))


if __name__ == "__main__":
    main()
