"""Cost of a GRU training step (BASELINE config 4's recurrence under aesara.grad, golden plan
``gru_bptt_b1_f32``: forward Scan + gradient Scan with mit-mot accumulators) at config 4's shape.
Both Scans run on the launch list / hipGraph (the persistent kernel covers forward recurrences
whose states are the only outputs).  usage: tools/bptt_probe.py [T H B]"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from golden_util import CASES, case_plan
from aesara_amd.executor import PlanExecutor
T, H, B = (int(v) for v in (sys.argv[1:4] or (512, 1024, 1)))
plan = case_plan(next(c for c in CASES if c["name"] == ("gru_bptt_b1_f32" if B == 1 else "gru_bptt_b4_f32")))
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(*((T, H) if B == 1 else (T, B, H)), device="cuda", generator=g) * 0.1
h0 = torch.zeros(*((H,) if B == 1 else (B, H)), device="cuda")
Ws = [torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
for mode in (False, True):
    ex = PlanExecutor(plan, use_graph=mode, borrow=True)
    for _ in range(3): outs = ex(x, h0, *Ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    n = 5
    for _ in range(n): outs = ex(x, h0, *Ws)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"config": "GRU BPTT T=%d H=%d fp32 B=%d (loss + 7 gradients)" % (T, H, B), "replay": mode,
                      "dev_ms": e0.elapsed_time(e1) / n, "wall_ms": (time.perf_counter() - t0) / n * 1e3,
                      "us_per_step": e0.elapsed_time(e1) / n / T * 1e3, "scan_modes": ex.scan_modes,
                      "finite": bool(all(torch.isfinite(o).all() for o in outs))}))
