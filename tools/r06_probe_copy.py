#!/usr/bin/env python
"""Big copies (DeepCopyOp of a contiguous 256 MiB matrix, Join of two of them) cold, with / without
streaming loads in csrc/copy.hip (AESARA_HIP_COPY_STREAM_BYTES)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch
from perf_probe import timeit, randn
from aesara_amd.plan import Node, Plan, Var
from aesara_amd.executor import PlanExecutor
f32 = torch.float32
def plan_n(op, in_vars, out_var, params):
    vs = {i: Var(i, dt, list(sh)) for i, (dt, sh) in enumerate(in_vars + [out_var])}
    n = len(in_vars)
    return Plan("c", vs, list(range(n)), [n], [Node(op, list(range(n)), [n], params)])
xs = [randn((16384, 4096), f32, k) for k in range(5)]          # 5 x 256 MiB: rotating = cold
nb = xs[0].numel() * 4
ex = PlanExecutor(plan_n("DeepCopyOp", [("float32", [None, None])], ("float32", [None, None]), {}), use_graph=True)
st = {"i": 0}
def call():
    st["i"] += 1
    ex(xs[st["i"] % 5])
d, w = timeit(call, 40, warmup=6)
print(json.dumps({"config": "DeepCopyOp f32 16384x4096 (rotating: cold)", "dev_us": d * 1e3, "GBs": 2 * nb / d / 1e6,
                  "frac": 2 * nb / d / 1e6 / 8000, "tune": os.environ.get("AESARA_HIP_COPY_STREAM_BYTES")}))
exj = PlanExecutor(plan_n("Join", [("int64", []), ("float32", [None, None]), ("float32", [None, None])],
                          ("float32", [None, None]), {}), use_graph=True)
def callj():
    st["i"] += 1
    exj(np.int64(0), xs[st["i"] % 5], xs[(st["i"] + 1) % 5])
d, w = timeit(callj, 30, warmup=6)
print(json.dumps({"config": "Join axis 0 of two f32 16384x4096 (rotating: cold)", "dev_us": d * 1e3, "GBs": 4 * nb / d / 1e6,
                  "frac": 4 * nb / d / 1e6 / 8000, "tune": os.environ.get("AESARA_HIP_COPY_STREAM_BYTES")}))
