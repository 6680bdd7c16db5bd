"""Fused-gate LSTM (H = 64, the width baked into the golden plans' gate slices) at a realistic T
and B: forward (golden lstm_fused_fwd_f32: persistent kernel) and one training step (golden
lstm_fused_bptt_h64_f32: forward persistent, gradient Scan on the launch list).
usage: tools/lstm_probe.py [T B D]"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from golden_util import CASES, case_plan
from aesara_amd.executor import PlanExecutor
T, B, D = (int(v) for v in (sys.argv[1:4] or (512, 64, 64)))
H = 64
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(T, B, D, device="cuda", generator=g) * 0.3
h0 = torch.zeros(B, H, device="cuda"); c0 = torch.zeros(B, H, device="cuda")
W = torch.randn(D, 4 * H, device="cuda", generator=g) / np.sqrt(D)
U = torch.randn(H, 4 * H, device="cuda", generator=g) / np.sqrt(H)
b = torch.zeros(4 * H, device="cuda")
for name in ("lstm_fused_fwd_f32", "lstm_fused_bptt_h64_f32"):
    plan = case_plan(next(c for c in CASES if c["name"] == name))
    for mode in (False, True):
        ex = PlanExecutor(plan, use_graph=mode, borrow=True)
        for _ in range(3): outs = ex(x, h0, c0, W, U, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): outs = ex(x, h0, c0, W, U, b)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"config": "%s T=%d B=%d D=%d H=64" % (name, T, B, D), "replay": mode, "dev_ms": ms,
                          "us_per_step": ms / T * 1e3, "scan_modes": ex.scan_modes}))
