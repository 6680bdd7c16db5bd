import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from golden_util import CASES, case_plan, case_inputs
from aesara_amd.executor import PlanExecutor
for name in ("nll_classifier_float32", "mlp_layers_float32", "cfg2_gauss_sum", "softmax_rows_f32", "hierarchical_logp_and_grad", "lstm_bptt_float32"):
    c = next(c for c in CASES if c["name"]==name)
    ins = [torch.from_numpy(np.ascontiguousarray(x)).cuda() if np.asarray(x).ndim else np.asarray(x) for x in case_inputs(c)]
    for mode in (False, True):
        ex = PlanExecutor(case_plan(c), use_graph=mode)
        for _ in range(5): ex(*ins)
        torch.cuda.synchronize(); t=time.perf_counter()
        n = 200 if name != "lstm_bptt_float32" else 30
        for _ in range(n): ex(*ins)
        torch.cuda.synchronize()
        print("%-28s %-6s %8.1f us/call (%d steps)" % (name, "replay" if mode else "eager", (time.perf_counter()-t)/n*1e6, len(ex.steps)))
