"""Warm the in-tree generated-kernel cache (aesara_amd/_kcache) without a GPU.

hiprtc cross-compiles for gfx950 on a machine with no device, so the fused kernels that the
golden plans need can be generated at build time; the GPU box then only loads code objects.
Kernel *selection* depends on run-time shapes/strides, so this replays the executor's host
logic with a recording stub in place of launches (no compute happens here).
"""
from __future__ import annotations

import json
import os


def prebuild_golden_kernels(limit=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = os.path.join(root, "tests", "golden", "cases.json")
    if not os.path.exists(cases):
        return 0
    import sys

    from .executor import PlanExecutor
    from .plan import Plan

    tests = os.path.join(root, "tests")
    if tests not in sys.path:
        sys.path.insert(0, tests)
    from golden_inputs import make_input

    with open(cases) as f:
        data = json.load(f)["cases"]
    n = 0
    for c in data[:limit]:
        ex = PlanExecutor(Plan.from_json(c["plan"]), dry_run=True)
        try:
            ex(*[make_input(s) for s in c["inputs"]])
            n += 1
        except Exception as e:  # a dry run cannot follow data-dependent control flow
            print(f"prebuild: {c['name']}: {type(e).__name__}: {e}")
    return n
