"""Round 6 probe: gradient.jacobian's row loop (golden plan scan_map_jacobian_rows: y = tanh(W x) * sum(x),
J = dy/dx) as ONE evaluation over whole sequences (fusion.batch_map_step) vs the step loop on the launch
list / hipGraph, and aesara.map over matrix rows (scan_map_rows_reduce_broadcast).  usage: tools/r06_probe_jacobian.py"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from golden_util import CASES, case_plan
from aesara_amd import executor as E

plan_of = lambda n: case_plan(next(c for c in CASES if c["name"] == n))  # noqa: E731


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) / n * 1e3


rng = np.random.default_rng(1)
for n in (256, 1024, 2048):
    x = torch.from_numpy(rng.standard_normal(n)).cuda()
    W = torch.from_numpy(rng.standard_normal((n, n)) / np.sqrt(n)).cuda()
    row = {"config": "jacobian of tanh(W x) * sum(x), n = %d (float64, %d steps)" % (n, n)}
    outs = {}
    for label, persist, graph in (("all_rows", 1, False), ("launch_list_eager", 0, False), ("launch_list_hipgraph", 0, True)):
        E.TUNE["scan_persist"] = persist
        try:
            ex = E.PlanExecutor(plan_of("scan_map_jacobian_rows"), use_graph=graph, borrow=True)
            d, w = timeit(lambda: ex(x, W), 3 if n > 256 else 5)
            outs[label] = ex(x, W)[0].clone()
            row[label + "_ms"] = [round(d, 3), round(w, 3)]
            row[label + "_mode"] = sorted(set(ex.scan_modes.values()))
        finally:
            E.TUNE["scan_persist"] = 1
    row["max_abs_diff_vs_launch_list"] = float((outs["all_rows"] - outs["launch_list_eager"]).abs().max())
    print(json.dumps(row))
M = torch.from_numpy(rng.standard_normal((65536, 256))).cuda()
b = torch.from_numpy(rng.standard_normal(256)).cuda()
idx = torch.from_numpy(rng.integers(-65536, 65536, 65536)).cuda()
row = {"config": "aesara.map over 65536 rows of 256 (softmax-like row, outer-product sum, gather, row sum; float64)"}
for label, persist in (("all_rows", 1), ("launch_list_eager", 0)):
    E.TUNE["scan_persist"] = persist
    try:
        ex = E.PlanExecutor(plan_of("scan_map_rows_reduce_broadcast"), use_graph=False, borrow=True)
        if persist:
            d, w = timeit(lambda: ex(M, b, idx), 3)
        else:       # (65536 steps of ~8 launches: one call is enough)
            torch.cuda.synchronize(); t0 = time.perf_counter(); ex(M, b, idx); torch.cuda.synchronize()
            d = w = (time.perf_counter() - t0) * 1e3
        row[label + "_ms"] = [round(d, 3), round(w, 3)]
    finally:
        E.TUNE["scan_persist"] = 1
print(json.dumps(row))
