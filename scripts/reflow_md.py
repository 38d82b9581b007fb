#!/usr/bin/env python3
"""Re-wraps the PROSE of a Markdown file to a line width (default 120): paragraphs and list items are re-filled, tables, fenced code,
headings, HTML and indented code stay as they are (a table row cannot be wrapped; its cells should be short).
    python scripts/reflow_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def reflow(text, width):
    out, para, indent, first = [], [], "", ""
    fence = False

    def flush():
        nonlocal para, indent, first
        if para:
            body = " ".join(s.strip() for s in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=first, subsequent_indent=indent, break_long_words=False,
                                     break_on_hyphens=False) or [first.rstrip()])
        para, indent, first = [], "", ""

    for line in text.split("\n"):
        s = line.rstrip()
        if s.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(s)
            continue
        if fence or not s.strip() or s.lstrip().startswith(("|", "#", "<", ">")) or re.match(r"^( {4,}|\t)", s) and not para:
            flush()
            out.append(s)
            continue
        m = re.match(r"^(\s*)([-*+]|\d+[.)])\s+", s)
        if m:  # a new list item
            flush()
            first = s[: m.end()]
            indent = " " * len(first)
            para = [s[m.end():]]
            continue
        if not para:
            lead = re.match(r"^\s*", s).group(0)
            first = indent = lead
        para.append(s)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    src = open(path).read()
    dst = reflow(src, width)
    open(path, "w").write(dst if dst.endswith("\n") else dst + "\n")
    long_lines = sum(1 for ln in dst.split("\n") if len(ln) > width)
    print(f"{path}: {len(src.splitlines())} -> {len(dst.splitlines())} lines, {long_lines} still longer than {width} (tables / code)")
