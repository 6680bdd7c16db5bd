#!/usr/bin/env python
"""Per-plan-step device time of one golden plan at a chosen shape (the executor's own profile:
HIP events around every step of an eager evaluation — what ``fn.profile`` is filled from).

usage (GPU box): python tools/step_profile.py [--case gru_bptt_b4_f32] [--T 512 --H 1024 --B 64]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="gru_bptt_b4_f32")
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--H", type=int, default=1024)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--evals", type=int, default=5)
    a = ap.parse_args()
    import numpy as np
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd.executor import PlanExecutor
    plan = case_plan(next(c for c in CASES if c["name"] == a.case))
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    T, H, B = a.T, a.H, a.B
    Ws = [torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    x = torch.randn(T, B, H, device="cuda", generator=g) * 0.1
    h0 = torch.zeros(B, H, device="cuda")
    ex = PlanExecutor(plan, use_graph=False)
    for _ in range(2):
        ex(x, h0, *Ws)
    subs = [(ex, "")]
    for inner, pre, _seqdots, lifted in ex._inner.values():      # the work a Scan step does around its loop
        if lifted is not None:
            subs.append((lifted["exec"], "  [sequence-only work of a Scan, up front] "))
        if pre is not None:
            subs.append((pre, "  [loop-invariant work of a Scan] "))
    for e, _tag in subs:
        e.enable_profile()
    for _ in range(a.evals):
        ex(x, h0, *Ws)
    rows = []
    for e, tag in subs:
        e._collect_profile()
        for si in range(len(e.steps)):
            if e.step_count[si]:
                rows.append({"step": tag + e.step_label(si), "ms": 1e3 * e.step_time[si] / a.evals})
    rows.sort(key=lambda r: -r["ms"])
    print(json.dumps({"case": a.case, "T": T, "H": H, "B": B, "scan_modes": ex.scan_modes,
                      "scan_notes": getattr(ex, "scan_notes", {}),
                      "total_ms": sum(r["ms"] for r in rows if not r["step"].startswith(" ")), "steps": rows}, indent=1))


if __name__ == "__main__":
    main()
