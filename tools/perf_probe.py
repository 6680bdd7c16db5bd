#!/usr/bin/env python
"""Per-config timing probe (MI355X): runs the BASELINE configs through the HIP linker at their
full shapes and reports device time per eval (HIP events on the launch stream), evals/s and
the roofline fraction of SURVEY §8(d).  Writes JSON lines; used to fill DESIGN.md's table."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

from golden_util import CASES, case_plan
from aesara_amd._lib import check, lib
from aesara_amd.executor import PlanExecutor


def plan_of(name):
    return case_plan(next(c for c in CASES if c["name"] == name))


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    e0, e1 = C.c_void_p(), C.c_void_p()
    check(lib.ahip_event_create(C.byref(e0)))
    check(lib.ahip_event_create(C.byref(e1)))
    t0 = time.perf_counter()
    check(lib.ahip_event_record(e0, stream))
    for _ in range(iters):
        fn()
    check(lib.ahip_event_record(e1, stream))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    ms = C.c_float()
    check(lib.ahip_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / iters, wall * 1e3


def randn(shape, dtype, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randn(*shape, dtype=dtype, device="cuda", generator=g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="gpurun_out/perf.jsonl")
    ap.add_argument("--graph", type=int, default=1)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    rows = []

    def report(name, dev_ms, wall_ms, work, unit, peak, **extra):
        ach = work / (dev_ms * 1e-3) / 1e9 if unit == "GB/s" else work / (dev_ms * 1e-3) / 1e12
        r = dict(tune={k: v for k, v in os.environ.items() if k.startswith("AESARA_HIP_")},
                 config=name, dev_ms=dev_ms, wall_ms=wall_ms, evals_per_s=1e3 / max(wall_ms, dev_ms),
                 achieved=ach, unit=unit, peak=peak, frac=ach / peak, **extra)
        rows.append(r)
        print(json.dumps(r), flush=True)

    want = lambda n: (not args.only) or any(t in n for t in args.only.split(","))  # noqa: E731
    f64, f32 = torch.float64, torch.float32
    G = bool(args.graph)

    if want("cfg1b"):
        ex = PlanExecutor(plan_of("cfg1b_matrix_add"), use_graph=G, borrow=True)
        x, y = randn((4096, 4096), f64, 0), randn((4096, 4096), f64, 1)
        d, w = timeit(lambda: ex(x, y), 200)
        report("cfg1b add f64 4096^2", d, w, 3 * 4096 * 4096 * 8, "GB/s", 8000.0)

    if want("cfg2"):
        ex = PlanExecutor(plan_of("cfg2_gauss_sum"), use_graph=G, borrow=True)
        x = randn((4096, 4096), f64, 1)
        mu = torch.tensor(0.1, dtype=f64, device="cuda")
        sg = torch.tensor(1.3, dtype=f64, device="cuda")
        d, w = timeit(lambda: ex(x, mu, sg), 500)
        report("cfg2 fused exp-sum f64 4096^2", d, w, 4096 * 4096 * 8, "GB/s", 8000.0)
        exu = PlanExecutor(plan_of("cfg2_gauss_sum"), use_graph=G, fuse=False, borrow=True)
        d, w = timeit(lambda: exu(x, mu, sg), 100)
        report("cfg2 UNFUSED (reference node structure)", d, w, 4096 * 4096 * 8, "GB/s", 8000.0)

    if want("redsum"):
        from aesara_amd.plan import Node, Plan, Var
        pl = Plan("redsum", {0: Var(0, "float64", [None, None]), 1: Var(1, "float64", [])}, [0], [1],
                  [Node("CAReduce", [0], [1], {"scalar_op": "add", "axis": None,
                                               "acc_dtype": "float64"})])
        ex = PlanExecutor(pl, use_graph=G, borrow=True)
        x = randn((4096, 4096), f64, 1)
        d, w = timeit(lambda: ex(x), 500)
        report("pure sum f64 4096^2 (streaming-read ceiling)", d, w, 4096 * 4096 * 8, "GB/s", 8000.0)

    if want("axisred"):
        from aesara_amd.plan import Node, Plan, Var

        def red_plan(dt, nd, axis, op="add"):
            out_nd = nd - len(axis)
            return Plan("axisred", {0: Var(0, dt, [None] * nd), 1: Var(1, dt, [None] * out_nd)},
                        [0], [1], [Node("CAReduce", [0], [1], {
                            "scalar_op": op, "axis": list(axis),
                            "acc_dtype": "float64" if op == "add" else dt})])
        for dt, tdt, shape, axis, op in (
                ("float64", f64, (4096, 4096), (0,), "add"), ("float64", f64, (4096, 4096), (1,), "add"),
                ("float32", f32, (16384, 4096), (0,), "add"), ("float32", f32, (16384, 4096), (1,), "add"),
                ("float32", f32, (4194304, 8), (0,), "add"), ("float32", f32, (8, 4194304), (1,), "add"),
                ("float32", f32, (4194304, 8), (1,), "maximum"),
                ("float32", f32, (256, 512, 256), (1,), "add"), ("float32", f32, (256, 512, 256), (0, 2), "add")):
            ex = PlanExecutor(red_plan(dt, len(shape), axis, op), use_graph=G, borrow=True)
            x = randn(shape, tdt, 3)
            if os.environ.get("PROBE_ROTATE"):
                # MALL-cold: rotate over enough copies that none is still in the 256 MiB memory-side cache
                nrot = max(2, -(-(1 << 30) // (x.numel() * x.element_size())))
                xs = [x] + [x.clone() for _ in range(nrot - 1)]
                st = {"i": 0}

                def call():
                    st["i"] += 1
                    ex(xs[st["i"] % nrot])
                d, w = timeit(call, 60, warmup=nrot)
            else:
                d, w = timeit(lambda: ex(x), 50)
            report("%s axis=%s %s %s%s" % (op, axis, dt, "x".join(map(str, shape)),
                                           " (rotating: cold)" if os.environ.get("PROBE_ROTATE") else ""), d, w,
                   x.numel() * x.element_size(), "GB/s", 8000.0)

    if want("misc"):
        # secondary ops of the path at scale: which of them are far from the HBM roofline?
        from aesara_amd.plan import Node, Plan, Var
        from aesara_amd.device import DevArray

        def one(op, in_vars, out_var, params):
            vs = {i: Var(i, dt, list(sh)) for i, (dt, sh) in enumerate(in_vars + [out_var])}
            n = len(in_vars)
            return Plan("misc", vs, list(range(n)), [n], [Node(op, list(range(n)), [n], params)])

        def ew(opname, dts, shapes, odt):
            sc = {"n_in": len(dts), "nodes": [{"op": opname, "in": [["i", k] for k in range(len(dts))],
                                                "dtype": odt}], "out": [["t", 0]]}
            return one("Elemwise", list(zip(dts, shapes)), (odt, shapes[0]), {"scalar": sc})

        x = randn((8192, 4096), f32, 1)
        b, c = randn((1, 4096), f32, 2), randn((8192, 1), f32, 3)
        nb = x.numel() * 4
        for name, pl, ins_, byt in (
                ("ew x+b[None,:] f32 8192x4096", ew("add", ["float32"] * 2, [[None, None], [1, None]], "float32"), (x, b), 2 * nb),
                ("ew x*c[:,None] f32 8192x4096", ew("mul", ["float32"] * 2, [[None, None], [None, 1]], "float32"), (x, c), 2 * nb),
                ("cumsum axis=0 f32 8192x4096", one("CumOp", [("float32", [None, None])], ("float32", [None, None]), {"axis": 0, "mode": "add"}), (x,), 2 * nb),
                ("cumsum axis=1 f32 8192x4096", one("CumOp", [("float32", [None, None])], ("float32", [None, None]), {"axis": 1, "mode": "add"}), (x,), 2 * nb),
                ("cumsum flat f32 2^25", one("CumOp", [("float32", [None])], ("float32", [None]), {"axis": 0, "mode": "add"}), (x.view(-1),), 2 * nb),
                ("argmax axis=1 f32 8192x4096", one("Argmax", [("float32", [None, None])], ("int64", [None]), {"axis": [1]}), (x,), nb),
                ("argmax axis=0 f32 8192x4096", one("Argmax", [("float32", [None, None])], ("int64", [None]), {"axis": [0]}), (x,), nb),
        ):
            ex = PlanExecutor(pl, use_graph=G, borrow=True)
            d, w = timeit(lambda: ex(*ins_), 10)
            report(name, d, w, byt, "GB/s", 8000.0)
        exs = PlanExecutor(ew("mul", ["float32"] * 2, [[None, None], [1, 1]], "float32"), use_graph=G, borrow=True)
        two = torch.full((1, 1), 2.0, dtype=f32, device="cuda")
        xs = DevArray.from_torch(x).view([4096, 2048], [8192, 2])
        d, w = timeit(lambda: exs(xs, two), 10)
        report("ew x[::2, ::2]*2 f32 (4096x2048 of 8192x4096)", d, w, 2 * 4096 * 2048 * 4, "GB/s", 8000.0)
        idx = torch.randint(0, 8192, (65536,), device="cuda")
        ext = PlanExecutor(one("AdvancedSubtensor1", [("float32", [None, None]), ("int64", [None])],
                               ("float32", [None, None]), {}), use_graph=G, borrow=True)
        d, w = timeit(lambda: ext(x, idx), 10)
        report("take 65536 rows of 4096 f32", d, w, 2 * 65536 * 4096 * 4, "GB/s", 8000.0)

    if want("misc2"):
        from aesara_amd.plan import Node, Plan, Var

        def plan_n(op, in_vars, out_var, params):
            vs = {i: Var(i, dt, list(sh)) for i, (dt, sh) in enumerate(in_vars + [out_var])}
            n = len(in_vars)
            return Plan("misc2", vs, list(range(n)), [n], [Node(op, list(range(n)), [n], params)])

        def ew1(opname, idt, odt):
            sc = {"n_in": 1, "nodes": [{"op": opname, "in": [["i", 0]], "dtype": odt}], "out": [["t", 0]]}
            return plan_n("Elemwise", [(idt, [None, None])], (odt, [None, None]), {"scalar": sc})
        x = randn((8192, 4096), f32, 1)
        nb = x.numel() * 4
        x8 = (x * 40).to(torch.int8)
        sc_gt = {"n_in": 2, "nodes": [{"op": "gt", "in": [["i", 0], ["i", 1]], "dtype": "bool"}], "out": [["t", 0]]}
        for name, pl, ins_, byt in (
                ("cast f32->f64 8192x4096", ew1("cast", "float32", "float64"), (x,), 3 * nb),
                ("neg int8 8192x4096", ew1("neg", "int8", "int8"), (x8,), 2 * x8.numel()),
                ("gt(int8,int8)->bool 8192x4096", plan_n("Elemwise", [("int8", [None, None])] * 2, ("bool", [None, None]), {"scalar": sc_gt}), (x8, x8), 3 * x8.numel()),
                ("exp f32 1024 (latency)", ew1("exp", "float32", "float32"), (x[:1, :1024],), 2 * 4096),
                ("join axis=0 2x(8192x4096 f32)", plan_n("Join", [("int64", []), ("float32", [None, None]), ("float32", [None, None])], ("float32", [None, None]), {}), (np.int64(0), x, x), 4 * nb),
                ("join axis=1 2x(8192x4096 f32)", plan_n("Join", [("int64", []), ("float32", [None, None]), ("float32", [None, None])], ("float32", [None, None]), {}), (np.int64(1), x, x), 4 * nb),
                ("alloc scalar -> 8192x4096 f32", plan_n("Alloc", [("float32", []), ("int64", []), ("int64", [])], ("float32", [None, None]), {}), (x[0, 0], np.int64(8192), np.int64(4096)), nb),
        ):
            ex = PlanExecutor(pl, use_graph=G, borrow=True)
            d, w = timeit(lambda: ex(*ins_), 10)
            report(name, d, w, byt, "GB/s", 8000.0)
        idx = torch.randint(0, 8192, (65536,), device="cuda")
        y = randn((65536, 1024), f32, 2)
        z = randn((8192, 1024), f32, 3)
        exs = PlanExecutor(plan_n("AdvancedIncSubtensor1", [("float32", [None, None]), ("float32", [None, None]), ("int64", [None])],
                                  ("float32", [None, None]), {"set_instead_of_inc": False, "inplace": False}), use_graph=G, borrow=True)
        d, w = timeit(lambda: exs(z, y, idx), 10)
        report("scatter-add 65536 rows of 1024 f32 into 8192 rows", d, w, 2 * y.numel() * 4, "GB/s", 8000.0)

    if want("gemmshapes"):
        # Dot22 / BatchedDot / Ger over shapes away from the square headline case
        from aesara_amd.plan import Node, Plan, Var

        def dot_plan(op, dt, nd):
            vs = {i: Var(i, dt, [None] * nd) for i in range(3)}
            return Plan("dot", vs, [0, 1], [2], [Node(op, [0, 1], [2], {})])
        for dt, tdt, pk in (("float32", f32, 157.3), ("float64", f64, 78.6)):
            exd = PlanExecutor(dot_plan("Dot22", dt, 2), use_graph=G, borrow=True)
            for M, N, K in ((8192, 8192, 512), (512, 512, 65536), (16384, 64, 1024), (64, 16384, 1024),
                            (2048, 2048, 2048), (4096, 4096, 64), (4000, 4000, 4000), (1024, 1024, 1024),
                            (65536, 256, 256), (256, 256, 256)):
                A, B = randn((M, K), tdt, 1), randn((K, N), tdt, 2)
                d, w = timeit(lambda: exd(A, B), 10)
                report("dot22 %s %dx%dx%d" % (dt, M, N, K), d, w, 2 * M * N * K, "TFLOP/s", pk)
            exb = PlanExecutor(dot_plan("BatchedDot", dt, 3), use_graph=G, borrow=True)
            for Bn, M, N, K in ((64, 512, 512, 512), (1024, 64, 64, 64), (16, 2048, 128, 2048)):
                A, B = randn((Bn, M, K), tdt, 1), randn((Bn, K, N), tdt, 2)
                d, w = timeit(lambda: exb(A, B), 10)
                report("batched_dot %s %dx(%dx%dx%d)" % (dt, Bn, M, N, K), d, w, 2 * Bn * M * N * K,
                       "TFLOP/s", pk)

    if want("cfg3a"):
        ex = PlanExecutor(plan_of("gemv_small_float64"), use_graph=G, borrow=True)
        M = randn((4096, 4096), f64, 2)
        v = randn((4096,), f64, 3)
        y = randn((4096,), f64, 4)
        d, w = timeit(lambda: ex(y, M, v), 200)
        report("cfg3a gemv f64 4096^2", d, w, 4096 * 4096 * 8 + 2 * 4096 * 8, "GB/s", 8000.0)
        ext = PlanExecutor(plan_of("gemv_T_float64"), use_graph=G, borrow=True)
        Mt = M.t()
        d, w = timeit(lambda: ext(y, Mt, v), 200)
        report("gemv f64 4096^2 (A.T view)", d, w, 4096 * 4096 * 8 + 2 * 4096 * 8, "GB/s", 8000.0)

    if args.only == "nn32":
        ex = PlanExecutor(plan_of("cfg3b_gemm_update"), use_graph=G, borrow=True)
        Cm = torch.zeros(4096, 4096, dtype=f32, device="cuda")
        A, B = randn((4096, 4096), f32, 3), randn((4096, 4096), f32, 4)
        d, w = timeit(lambda: ex(Cm, A, B), 30)
        report("cfg3b gemm f32 4096^3", d, w, 2 * 4096 ** 3, "TFLOP/s", 157.3)
        At, Bt = A.t().contiguous().t(), B.t().contiguous().t()
        d, w = timeit(lambda: ex(Cm, A, Bt), 30)
        report("gemm f32 4096^3 NT(B k-contig)", d, w, 2 * 4096 ** 3, "TFLOP/s", 157.3)

    if want("cfg3b"):
        ex = PlanExecutor(plan_of("cfg3b_gemm_update"), use_graph=G, borrow=True)
        Cm = torch.zeros(4096, 4096, dtype=f32, device="cuda")
        A, B = randn((4096, 4096), f32, 3), randn((4096, 4096), f32, 4)
        d, w = timeit(lambda: ex(Cm, A, B), 30)
        report("cfg3b gemm f32 4096^3", d, w, 2 * 4096 ** 3, "TFLOP/s", 157.3)
        for name, a, b in (("NT", A, B.t()), ("TN", A.t(), B), ("TT", A.t(), B.t())):
            d, w = timeit(lambda: ex(Cm, a, b), 20)
            report(f"gemm f32 4096^3 {name}", d, w, 2 * 4096 ** 3, "TFLOP/s", 157.3)
        ex64 = PlanExecutor(plan_of("gemm_T000_float64") if any(
            c["name"] == "gemm_T000_float64" for c in CASES) else plan_of("gemm1_float64"),
            use_graph=G, borrow=True)
        C64 = torch.zeros(4096, 4096, dtype=f64, device="cuda")
        A64, B64 = randn((4096, 4096), f64, 3), randn((4096, 4096), f64, 4)
        d, w = timeit(lambda: ex64(C64, A64, B64), 10)
        report("gemm f64 4096^3", d, w, 2 * 4096 ** 3, "TFLOP/s", 78.6)

    if want("cfg4") or args.only == "gruB1":
        T, H = int(os.environ.get("AESARA_PROBE_T", 512)), 1024
        ex = PlanExecutor(plan_of("cfg4_gru_b1_f32"), use_graph=G, borrow=True)
        x = randn((T, H), f32, 4) * 0.1
        h0 = torch.zeros(H, dtype=f32, device="cuda")
        Ws = [randn((H, H), f32, 5 + k) / np.sqrt(H) for k in range(6)]
        t0 = time.perf_counter()
        ex(x, h0, *Ws)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        d, w = timeit(lambda: ex(x, h0, *Ws), 5 if T > 64 else 1, warmup=1)
        report("cfg4 scan GRU T=%d H=1024 f32 B=1" % T, d, w, T * 6 * H * H * 4, "GB/s", 8000.0,
               us_per_step=d * 1e3 / T, first_call_s=first, scan_modes=list(ex.scan_modes.values()))

    if want("softmax"):
        Nr, Kc = 1 << 16, 1024
        x = randn((Nr, Kc), f32, 11) * 3
        for label, kw in (("softmax rows f32 65536x1024 (row-chain kernel)", {}),
                          ("softmax rows f32 65536x1024 UNFUSED (3 passes)", {"fuse": False})):
            ex = PlanExecutor(plan_of("softmax_rows_f32"), use_graph=G, borrow=True, **kw)
            d, w = timeit(lambda: ex(x), 20, warmup=3)
            report(label, d, w, 2 * Nr * Kc * 4, "GB/s", 8000.0)
        xs = randn((1 << 20, 64), f32, 12)
        ex = PlanExecutor(plan_of("softmax_rows_f32"), use_graph=G, borrow=True)
        d, w = timeit(lambda: ex(xs), 20, warmup=3)
        report("softmax rows f32 1048576x64 (16 rows per wave)", d, w, 2 * (1 << 20) * 64 * 4,
               "GB/s", 8000.0)

    if want("vocab"):
        Nr, Kc = 8192, 50304
        x = randn((Nr, Kc), f32, 31) * 3
        for label, kw in (("softmax rows f32 8192x50304 (long-row chain kernel)", {}),
                          ("softmax rows f32 8192x50304 UNFUSED (3 passes)", {"fuse": False})):
            ex = PlanExecutor(plan_of("softmax_rows_f32"), use_graph=G, borrow=True, **kw)
            d, w = timeit(lambda: ex(x), 10, warmup=2)
            report(label, d, w, 2 * Nr * Kc * 4, "GB/s", 8000.0)

    if want("layernorm"):
        x = randn((64, 1024, 1024), f32, 13)
        g, b = randn((1024,), f32, 14), randn((1024,), f32, 15)
        for label, kw in (("layernorm f32 64x1024x1024 (row-chain kernel)", {}),
                          ("layernorm f32 64x1024x1024 UNFUSED", {"fuse": False})):
            ex = PlanExecutor(plan_of("layernorm_float32"), use_graph=G, borrow=True, **kw)
            d, w = timeit(lambda: ex(x, g, b), 20, warmup=3)
            report(label, d, w, 2 * x.numel() * 4, "GB/s", 8000.0)

    if want("cfg4b64"):
        T, H, B = 512, 1024, 64
        ex = PlanExecutor(plan_of("cfg4_gru_b8_f32"), use_graph=G, borrow=True)
        x = randn((T, B, H), f32, 4) * 0.1
        h0 = torch.zeros((B, H), dtype=f32, device="cuda")
        Ws = [randn((H, H), f32, 5 + k) / np.sqrt(H) for k in range(6)]
        ex(x, h0, *Ws)
        torch.cuda.synchronize()
        d, w = timeit(lambda: ex(x, h0, *Ws), 3, warmup=1)
        report("cfg4 scan GRU T=512 H=1024 f32 B=64 (matrix state)", d, w,
               T * 6 * 2 * B * H * H, "TFLOP/s", 157.3, us_per_step=d * 1e3 / T)

    if want("gemmsmall"):
        from aesara_amd._lib import lib as _l
        for (M, N, K) in ((64, 1024, 1024), (8, 1024, 1024), (256, 1024, 1024), (64, 4096, 4096),
                          (512, 512, 512), (1024, 1024, 1024), (2048, 1024, 1024)):
            Cm = torch.zeros((M, N), dtype=f32, device="cuda")
            A, B = randn((M, K), f32, 3), randn((K, N), f32, 4)
            for tiles in (1, 100000):
                check(_l.ahip_set_param(b"gemm_small_max_tiles", tiles))
                exx = PlanExecutor(plan_of("cfg3b_gemm_update"), use_graph=G, borrow=True)
                d, w = timeit(lambda: exx(Cm, A, B), 20, warmup=3)
                report("gemm f32 %dx%dx%d %s" % (M, N, K, "big-tile" if tiles == 1 else "small-tile"),
                       d, w, 2 * M * N * K, "TFLOP/s", 157.3)
        check(_l.ahip_set_param(b"gemm_small_max_tiles", 64))

    if want("nll"):
        Nb, Dd, Cc = 32768, 1024, 1000
        ex = PlanExecutor(plan_of("nll_classifier_float32"), use_graph=G, borrow=True)
        x = randn((Nb, Dd), f32, 21)
        W = randn((Dd, Cc), f32, 22) * 0.03
        b = randn((Cc,), f32, 23) * 0.1
        y = torch.randint(0, Cc, (Nb,), device="cuda")
        d, w = timeit(lambda: ex(x, W, b, y), 10, warmup=2)
        report("softmax-classifier NLL + grads f32 N=32768 D=1024 C=1000", d, w,
               3 * 2 * Nb * Dd * Cc, "TFLOP/s", 157.3,
               note="flops = forward GEMM + dW GEMM + dX-free; includes log-softmax, gather, scatter")

    if want("transposed"):
        ex = PlanExecutor(plan_of("cfg1b_matrix_add"), use_graph=G, borrow=True)
        x, y = randn((4096, 4096), f64, 0), randn((4096, 4096), f64, 1)
        d, w = timeit(lambda: ex(x, y.t()), 20)
        report("cfg1b add f64 4096^2 with y.T view (strided operand)", d, w, 3 * x.numel() * 8,
               "GB/s", 8000.0)
        from aesara_amd.device import DevArray
        exi = PlanExecutor(plan_of("cfg1b_matrix_add"))
        v = DevArray.from_torch(y).view([4096, 4096], [1, 4096])
        d, w = timeit(lambda: exi.materialize(v), 20)
        report("materialise transpose f64 4096^2 (strided copy)", d, w, 2 * x.numel() * 8,
               "GB/s", 8000.0)

        exr = PlanExecutor(plan_of("reduce_all_transposed_float64"), use_graph=G, borrow=True)
        d, w = timeit(lambda: exr(x, y), 20)
        report("(x*y.T).sum(), max(y.T-x), sum(sqr(x.T)+y) f64 4096^2 (3 tiled reduces)", d, w,
               3 * 2 * x.numel() * 8, "GB/s", 8000.0)
        exf = PlanExecutor(plan_of("ew_transposed_float32_64x128"), use_graph=G, borrow=True)
        xf, yf, vf = randn((8192, 4096), f32, 0), randn((4096, 8192), f32, 1), randn((4096,), f32, 2)
        d, w = timeit(lambda: exf(xf, yf, vf), 20)
        report("x+y.T, exp(y.T/4)*x-v, sqr(y.T) f32 8192x4096 (3 tiled kernels)", d, w,
               (3 + 3 + 2) * xf.numel() * 4, "GB/s", 8000.0)

    if want("bptt"):
        T, H = 512, 1024
        ex = PlanExecutor(plan_of("scan_grad_last_state_f32"), use_graph=G, borrow=True)
        x = randn((T, H), f32, 41) * 0.1
        h0 = torch.zeros(H, dtype=f32, device="cuda")
        W = randn((H, H), f32, 42) / np.sqrt(H)
        U = randn((H, H), f32, 43) / np.sqrt(H)
        t0 = time.perf_counter()
        ex(x, h0, W, U)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        d, w = timeit(lambda: ex(x, h0, W, U), 3, warmup=1)
        report("tanh-RNN forward + BPTT (grad of last state) T=512 H=1024 f32", d, w,
               T * 3 * H * H * 4, "GB/s", 8000.0, us_per_step=d * 1e3 / T, first_call_s=first)

    if want("cfg5"):
        N, D = 1 << 22, 256
        ex = PlanExecutor(plan_of("cfg5_logistic"), use_graph=G, borrow=True)
        X = randn((N, D), f32, 6)
        wv = randn((D,), f32, 7) / 16
        b = torch.tensor(0.1, dtype=f32, device="cuda")
        yv = (torch.rand(N, device="cuda") < 0.5).to(f32)
        d, w = timeit(lambda: ex(X, wv, b, yv), 10, warmup=2)
        report("cfg5 logistic logp+grad f32 N=2^22 D=256", d, w, N * D * 4 + N * 4, "GB/s", 8000.0,
               note="algorithmic bytes = X once + y (reference graph reads X twice)")

    with open(args.out, "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
