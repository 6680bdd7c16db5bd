#!/usr/bin/env python
"""One variant of the matrix-state persistent Scan kernel (BASELINE config 4, B = 64) — result
against a plain torch loop, device time per evaluation.  The variant comes from the
AESARA_HIP_SM_* switches in the environment (tools/sm_variants.sh runs a list of them).

usage (GPU box): AESARA_HIP_SM_INIT=publish python tools/sm_ab.py [--B 64] [--trace]
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
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--H", type=int, default=1024)
    ap.add_argument("--evals", type=int, default=5)
    a = ap.parse_args()
    import numpy as np
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd import knobs
    from aesara_amd.executor import PlanExecutor
    T, H, B = a.T, a.H, a.B
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    Ws = [torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    x = torch.randn(T, B, H, device="cuda", generator=g) * 0.1
    h0 = torch.randn(B, H, device="cuda", generator=g) * 0.1
    ex = PlanExecutor(case_plan(next(c for c in CASES if c["name"] == "cfg4_gru_b8_f32")), borrow=True)
    got = ex(x, h0, *Ws)[-1].clone()
    Wz, Uz, Wr, Ur, Wh, Uh = [w.double() for w in Ws]
    h = h0.double()
    for t in range(T):
        xt = x[t].double()
        z = torch.sigmoid(xt @ Wz + h @ Uz)
        r = torch.sigmoid(xt @ Wr + h @ Ur)
        hh = torch.tanh(xt @ Wh + (r * h) @ Uh)
        h = (1 - z) * h + z * hh
    err = float((got.double() - h).norm() / h.norm())
    for _ in range(2):
        ex(x, h0, *Ws)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.evals):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ex(x, h0, *Ws)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    flops = T * 6 * 2 * B * H * H
    print(json.dumps({"variant": {k: os.environ[k] for k in sorted(os.environ) if k.startswith("AESARA_HIP_SM_")},
                      "scan_modes": ex.scan_modes, "rel_err_vs_fp64_loop": err, "ms": round(ms, 4),
                      "ms_min": round(ts[0], 4), "frac_fp32_mfma_peak": round(flops / (ms * 1e-3) / 157.3e12, 4)}))


if __name__ == "__main__":
    main()
