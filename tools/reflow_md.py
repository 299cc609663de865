#!/usr/bin/env python3
"""Reflow the prose of a markdown file to <= 118 columns, paragraph by paragraph (tables, headings and code fences untouched).
   python tools/reflow_md.py DESIGN.md"""
import re
import sys
import textwrap

p = sys.argv[1]
lines = open(p).read().split("\n")
out, para = [], None


def flush():
    global para
    if para is None:
        return
    lead, cont, text = para
    w = textwrap.wrap(text, width=118 - len(cont), break_long_words=False, break_on_hyphens=False)
    out.append(lead + w[0])
    out.extend(cont + x for x in w[1:])
    para = None


fence = False
for line in lines:
    if line.startswith("```"):
        flush(); fence = not fence; out.append(line); continue
    if fence or line.startswith("|") or line.startswith("#") or line.strip() == "":
        flush(); out.append(line); continue
    m = re.match(r"^(\s*(?:[*]|\d+\.)\s+)", line)
    if m:
        flush(); lead = m.group(1); para = [lead, " " * len(lead), line[len(lead):].strip()]
    elif para is not None:
        para[2] += " " + line.strip()
    else:
        para = ["", "", line.strip()]
flush()
open(p, "w").write("\n".join(out))
