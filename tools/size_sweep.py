"""Fixed overhead of the fused Elemwise+Sum kernel (config 2): device time vs problem size."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tools')
import torch
from perf_probe import plan_of, timeit
from aesara_amd.executor import PlanExecutor
ex = PlanExecutor(plan_of("cfg2_gauss_sum"), use_graph=True)
mu = torch.tensor(0.1, dtype=torch.float64, device="cuda"); sg = torch.tensor(1.3, dtype=torch.float64, device="cuda")
for n in (1024, 2048, 2896, 4096, 5792, 8192, 16384):
    x = torch.randn((n, n), dtype=torch.float64, device="cuda")
    d, w = timeit(lambda: ex(x, mu, sg), 50, warmup=5)
    print("n=%5d  %8.1f MiB  %8.2f us  %7.1f GB/s" % (n, n * n * 8 / 2**20, d * 1e3, n * n * 8 / d / 1e6))
