#!/usr/bin/env python
"""The twelve GEMMs of the GRU training step (B = 64, T = 512, H = 1024: [32768, 1024] @ [1024, 1024] and the
weight-gradient form [1024, 32768] @ [32768, 1024]) in isolation: device time per layout / GROUP setting."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
from perf_probe import timeit, randn
from aesara_amd.plan import Node, Plan, Var
from aesara_amd.executor import PlanExecutor
f32 = torch.float32
vs = {i: Var(i, "float32", [None, None]) for i in range(3)}
ex = PlanExecutor(Plan("dot", vs, [0, 1], [2], [Node("Dot22", [0, 1], [2], {})]), use_graph=True, borrow=True)
X = randn((32768, 1024), f32, 1); W = randn((1024, 1024), f32, 2); D = randn((32768, 1024), f32, 3)
rows = []
for name, a, b, fl in (("x @ W  [32768,1024]@[1024,1024] NN", X, W, 2 * 32768 * 1024 * 1024),
                       ("d @ W.T [32768,1024]@[1024,1024]^T NT", D, W.t(), 2 * 32768 * 1024 * 1024),
                       ("x.T @ d [1024,32768]@[32768,1024] TN (weight gradient)", X.t(), D, 2 * 32768 * 1024 * 1024)):
    d, w = timeit(lambda: ex(a, b), 20)
    r = {"config": name, "tune": {k: v for k, v in os.environ.items() if k.startswith("AESARA_HIP_")},
         "dev_ms": d, "TFLOPs": fl / d / 1e9, "frac": fl / d / 1e9 / 157.3}
    print(json.dumps(r), flush=True)
