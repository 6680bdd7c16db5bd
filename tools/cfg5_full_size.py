"""BASELINE config 5 at its full shape (N = 2^24, D = 256 fp32: 16 GiB of X) on one MI355X: timing of
the single-pass row-program kernel and error against an fp64 restatement accumulated in row blocks."""
import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tools')
import torch
from golden_util import CASES, case_plan
from aesara_amd.executor import PlanExecutor
N, D = 1 << 24, 256
g = torch.Generator(device="cuda"); g.manual_seed(6)
X = torch.empty((N, D), dtype=torch.float32, device="cuda")
for i in range(0, N, 1 << 20):
    X[i:i + (1 << 20)] = torch.randn((1 << 20, D), dtype=torch.float32, device="cuda", generator=g)
w = torch.randn((D,), dtype=torch.float32, device="cuda", generator=g) / 16
b = torch.tensor(0.1, dtype=torch.float32, device="cuda")
y = (torch.rand(N, device="cuda", generator=g) < 0.5).float()
ex = PlanExecutor(case_plan(next(c for c in CASES if c["name"] == "cfg5_logistic")), use_graph=True)
outs = ex(X, w, b, y); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): outs = ex(X, w, b, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
logp, gw, gb = [o.clone() for o in outs]
# fp64 restatement in row blocks
ref_logp = 0.0; ref_gb = 0.0; ref_gw = torch.zeros(D, dtype=torch.float64, device="cuda")
for i in range(0, N, 1 << 21):
    Xd = X[i:i + (1 << 21)].double(); yd = y[i:i + (1 << 21)].double()
    z = Xd @ w.double() + 0.1
    ref_logp += -(yd * torch.nn.functional.softplus(-z) + (1 - yd) * torch.nn.functional.softplus(z)).sum().item()
    r = yd - torch.sigmoid(z); ref_gb += r.sum().item(); ref_gw += Xd.t() @ r
res = {"config": "cfg5 FULL SIZE N=2^24 D=256 f32 (16 GiB of X)", "ms": dt * 1e3,
       "GB/s": (N * D * 4 + N * 4) / dt / 1e9, "frac_of_8TBs": (N * D * 4 + N * 4) / dt / 8e12,
       "logp_rel_err": abs(logp.item() - ref_logp) / abs(ref_logp),
       "gb_abs_err": abs(gb.item() - ref_gb),
       "gw_rel_err": (torch.linalg.norm(gw.double() - ref_gw) / torch.linalg.norm(ref_gw)).item()}
print(json.dumps(res))
