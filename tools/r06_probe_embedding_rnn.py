"""Round 6 probe: an RNN whose step looks its input up itself (golden plan scan_embedding_lookup_batch_f32:
h_t = tanh(E[idx_t] + h_{t-1} U), cost = sum(h_T^2), outputs hs, dcost/dU, dcost/dE) at T = 512, B = 64,
H = 1024, a 50 000-row table.  default: lookup + table-gradient scatter taken out of the loops
(fusion.push_out_sequence_glue / push_out_product_accumulators), both Scans on the persistent matrix kernel;
launch list: the same plan with AESARA_HIP_SCAN_PERSIST=0; unfused: fuse=False (the graph as lowered: lookup and
scatter inside the step loops, one host read of the indices per step)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from golden_util import CASES, case_plan
from aesara_amd import executor as E

plan = case_plan(next(c for c in CASES if c["name"] == "scan_embedding_lookup_batch_f32"))
T, B, V, H = 512, 64, 50000, 1024
rng = np.random.default_rng(1)
idx = torch.from_numpy(rng.integers(0, V, (T, B))).cuda()
Em = torch.from_numpy((rng.standard_normal((V, H)) * 0.5).astype("float32")).cuda()
U = torch.from_numpy((rng.standard_normal((H, H)) / np.sqrt(H)).astype("float32")).cuda()
h0 = torch.from_numpy((rng.standard_normal((B, H)) * 0.1).astype("float32")).cuda()


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


row = {"config": "embedding RNN training step T=%d B=%d H=%d V=%d fp32 (hs, dU, dE)" % (T, B, H, V)}
ref = None
for label, persist, fuse, n in (("default", 1, True, 5), ("launch_list", 0, True, 2), ("unfused", 1, False, 1)):
    E.TUNE["scan_persist"] = persist
    try:
        ex = E.PlanExecutor(plan, use_graph=False, borrow=True, fuse=fuse)
        row[label + "_ms"] = round(timeit(lambda: ex(idx, Em, U, h0), n), 3)
        row[label + "_modes"] = sorted(set(ex.scan_modes.values()))
        outs = [o.clone() for o in ex(idx, Em, U, h0)]
        if ref is None:
            ref = outs
        else:
            row[label + "_max_rel_diff"] = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs, ref))
    finally:
        E.TUNE["scan_persist"] = 1
print(json.dumps(row))
