"""Round 6 probe: can throughput GEMMs share the chip with a persistent matrix-state Scan kernel?
("... or on a second stream ...; check what fits beside one wave per SIMD".)

The forward GRU Scan of BASELINE config 4 at B = 64 (golden plan cfg4_gru_b8_f32: one up-front
x @ Wh product + the persistent sm_* kernel, 256 workgroups x 4 waves with the weight columns in
VGPRs) on the current stream, three [T*B, H] @ [H, H] products (what the gradient Scan asks for in
front of its loop) on a second stream.  Reports: each alone, back to back on one stream, and both
streams together in either launch order."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from golden_util import CASES, case_plan
from aesara_amd.executor import PlanExecutor

T, H, B = 512, 1024, 64
plan_of = lambda n: case_plan(next(c for c in CASES if c["name"] == n))  # noqa: E731
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(T, B, H, device="cuda", generator=g) * 0.1
h0 = torch.zeros(B, H, device="cuda")
Ws = [torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
x2 = x.reshape(T * B, H)
ex_f = PlanExecutor(plan_of("cfg4_gru_b8_f32"), use_graph=False, borrow=True)
ex_g = [PlanExecutor(plan_of("dot22_f32"), use_graph=False, borrow=True) for _ in range(3)]
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def scan():
    return ex_f(x, h0, *Ws)


def gemms():
    return [e(x2, Ws[k]) for k, e in enumerate(ex_g)]


def both(order):
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    if order == "scan_first":
        a = scan()
        with torch.cuda.stream(side):
            b = gemms()
    else:
        with torch.cuda.stream(side):
            b = gemms()
        a = scan()
    ev2 = torch.cuda.Event()
    ev2.record(side)
    main.wait_event(ev2)
    return a, b


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.cuda.stream(side):
    gemms()
torch.cuda.synchronize()
ref_h = scan()[-1].clone()
ref_g = [o[0].clone() for o in gemms()]
rows = {}
for rnd in range(2):
    rows.setdefault("scan_alone_ms", []).append(timeit(scan))
    rows.setdefault("three_gemms_alone_ms", []).append(timeit(gemms))
    rows.setdefault("serial_one_stream_ms", []).append(timeit(lambda: (scan(), gemms())))
    for order in ("scan_first", "gemms_first"):
        rows.setdefault("two_streams_%s_ms" % order, []).append(timeit(lambda: both(order)))
a, b = both("scan_first")
torch.cuda.synchronize()
rows["same_results"] = bool(torch.equal(a[-1], ref_h) and all(torch.equal(o[0], r) for o, r in zip(b, ref_g)))
rows["scan_modes"] = list(ex_f.scan_modes.values())
print(json.dumps(rows))
