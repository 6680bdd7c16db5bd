# A/B: 64x64-tile GEMM (gemm_half_*) vs the default dispatch on mid-size fp32 / fp64 problems
cat > /tmp/gh.py <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
from golden_util import CASES, case_plan
from aesara_amd.executor import PlanExecutor
from aesara_amd._lib import lib, check
plan = case_plan(next(c for c in CASES if c["name"] == ("gemm0_float64" if os.environ.get("F64") else "cfg3b_gemm_update")))
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
shapes = [(512,512,512),(768,768,1024),(1024,1024,1024),(1024,1024,4096),(1536,1024,1024),(2048,1024,1024),(2048,2048,512),
          (2048,2048,2048),(3072,1024,1024),(4096,1024,1024),(1024,4096,1024),(3072,2048,1024),(4096,2048,1024),(4096,4096,1024), (1000,1000,1024), (2500, 1100, 512)]
for (M,N,K) in shapes:
    dt = torch.float64 if os.environ.get("F64") else torch.float32
    A = torch.randn(M, K, device="cuda", dtype=dt); B = torch.randn(K, N, device="cuda", dtype=dt); Cm = torch.zeros(M, N, device="cuda", dtype=dt)
    res = {}
    for label, mx in (("default", 1), ("half", 1 << 40)):
        check(lib.ahip_set_param(b"gemm_half_max_tiles", mx))
        ex = PlanExecutor(plan, use_graph=True, borrow=True)
        ms = timeit(lambda: ex(Cm, A, B))
        res[label] = ms
    t128 = -(-M//128) * -(-N//128)
    print("%5dx%5dx%5d tiles128=%4d default %.1f us (%.1f TF)  half %.1f us (%.1f TF)" % (M,N,K,t128, res["default"]*1e3, 2*M*N*K/res["default"]/1e9, res["half"]*1e3, 2*M*N*K/res["half"]/1e9))
PY
python /tmp/gh.py
