"""Is the NN layout of the 4096^3 fp32 Gemm really slower, or is it measured first?  Times the four
layouts in two different orders (HIP events, 20 evals each after 5 warm-ups)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from golden_util import CASES, case_plan
from aesara_amd.executor import PlanExecutor
plan = case_plan(next(c for c in CASES if c["name"] == "cfg3b_gemm_update"))
n = 4096
g = torch.Generator(device="cuda"); g.manual_seed(3)
A = torch.randn(n, n, device="cuda", generator=g); B = torch.randn(n, n, device="cuda", generator=g)
C = torch.zeros(n, n, device="cuda")
lay = {"NN": (A, B), "NT": (A, B.t().contiguous().t()), "TN": (A.t().contiguous().t(), B),
       "TT": (A.t().contiguous().t(), B.t().contiguous().t())}
def t(name):
    a, b = lay[name]
    ex = PlanExecutor(plan, use_graph=True, borrow=True)
    for _ in range(5): ex(C, a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ex(C, a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    return "%s %.4f ms %.1f TF" % (name, ms, 2 * n ** 3 / ms / 1e9)
for order in (["NN", "NT", "TN", "TT"], ["TT", "TN", "NT", "NN"], ["NN", "NN", "TN", "NN"]):
    print(" | ".join(t(x) for x in order))
