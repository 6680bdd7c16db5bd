#!/usr/bin/env python
"""Per-phase timeline of the VECTOR-state persistent Scan kernel (BASELINE config 4, B = 1: the
`sp_*` kernel): where each microsecond of a recurrent step goes (the kernel runs
at 1.89 x its measured hand-off floor and nothing said where the difference is).  Thread 0 of two
workgroups stamps s_memtime at every mark of 32 consecutive steps (AESARA_HIP_SP_TRACE=1 build of the
generated kernel); the cycle counter is calibrated against s_memrealtime (100 MHz) over the kernel.

usage (GPU box): python tools/sp_trace.py [--case cfg4_gru_b1_f32] > gpurun_out/sp_trace.json
"""
import argparse
import json
import os
import sys

os.environ["AESARA_HIP_SP_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="cfg4_gru_b1_f32")
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--H", type=int, default=1024)
    a = ap.parse_args()
    import numpy as np
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd import scan_persist as sp
    from aesara_amd.executor import PlanExecutor
    plan = case_plan(next(c for c in CASES if c["name"] == a.case))
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    T, H = a.T, a.H
    Ws = [torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    x = torch.randn(T, H, device="cuda", generator=g) * 0.1
    h0 = torch.zeros(H, device="cuda")
    ex = PlanExecutor(plan, use_graph=False)
    for _ in range(3):
        ex(x, h0, *Ws)
    torch.cuda.synchronize()
    ctl, marks = ex.sp_trace
    raw = ctl[16:].cpu().numpy().view(np.uint64)
    n = sp.TRACE_NT * sp.TRACE_MARKS
    out = {"case": a.case, "T": T, "H": H, "scan_modes": ex.scan_modes, "marks": marks,
           "traced_steps": [sp.TRACE_T0, sp.TRACE_T0 + sp.TRACE_NT],
           "switches": {k: os.environ[k] for k in sorted(os.environ) if k.startswith("AESARA_HIP_")},
           "workgroups": {}}
    for w, name in enumerate(("wg0", "wg_mid")):
        blk = raw[w * (n + 4):(w + 1) * (n + 4)]
        st = blk[:n].reshape(sp.TRACE_NT, sp.TRACE_MARKS)[:, :len(marks)].astype(np.int64)
        c0, r0, c1, r1 = [int(v) for v in blk[n:n + 4]]
        mhz = (c1 - c0) / ((r1 - r0) / 100.0)                   # cycles per microsecond
        d = np.diff(st, axis=1) / mhz                           # us between consecutive marks
        step = (st[1:, 0] - st[:-1, 0]) / mhz
        tail = (st[1:, 0] - st[:-1, -1]) / mhz                  # last mark -> next step's first
        out["workgroups"][name] = {
            "counter_MHz": mhz, "kernel_us": (r1 - r0) / 100.0, "us_per_step_mean": float(step.mean()),
            "us_per_step_min_max": [float(step.min()), float(step.max())],
            "segments_us_mean": {"%s -> %s" % (marks[i], marks[i + 1]): round(float(d[:, i].mean()), 3)
                                 for i in range(len(marks) - 1)},
            "segments_us_std": {"%s -> %s" % (marks[i], marks[i + 1]): round(float(d[:, i].std()), 3)
                                for i in range(len(marks) - 1)},
            "loop_back_us_mean": round(float(tail.mean()), 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
