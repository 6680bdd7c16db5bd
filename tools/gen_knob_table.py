#!/usr/bin/env python
"""Rewrite the switch table ``docs/SWITCHES.md`` (DESIGN.md §3.5 points at it) from the registry
``aesara_amd/knobs.py``."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aesara_amd import knobs  # noqa: E402

path = os.path.join(ROOT, "docs", "SWITCHES.md")
src = open(path).read()
head = "| switch | default | what |\n|---|---|---|\n"
i = src.index(head) + len(head)
j = i
while src[j:j + 1] == "|":
    j = src.index("\n", j) + 1
rows = "".join("| `%s` | %s | %s |\n" % (n, "—" if d is None else d, doc.replace("|", "/"))
               for n, d, _c, doc in knobs.table())
open(path, "w").write(src[:i] + rows + src[j:])
print("wrote %d switches" % len(knobs.table()))
