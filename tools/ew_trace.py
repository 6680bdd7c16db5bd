#!/usr/bin/env python
"""In-kernel timeline of the fused Elemwise + full-reduction kernel (BASELINE config 2, MALL-cold).

An AESARA_HIP_EW_TRACE=1 build of the generated kernel makes thread 0 of EVERY workgroup stamp the
100 MHz constant clock (s_memrealtime, comparable across XCDs) at: kernel entry, first group of
vectors evaluated, streaming loop done, partial published; workgroup 0 adds: all partials
collected, result stored.  The host reads the stamps after each of N cold evaluations and reports
where the microseconds of one launch go; the same run times the traced and the untraced kernel
with HIP events (the difference between the event time per eval and the in-kernel span is the
kernel boundary).

usage (GPU box): python tools/ew_trace.py [--rows 4096] [--evals 24] > gpurun_out/ew_trace.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def pct(a, q):
    import numpy as np
    return float(np.percentile(a, q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--evals", type=int, default=24)
    ap.add_argument("--sum-only", action="store_true", help="the read-only ceiling kernel (pure sum)")
    ap.add_argument("--raw", default="", help="save the raw stamps of the last 4 traced evals (npz)")
    a = ap.parse_args()
    os.environ["AESARA_HIP_EW_TRACE"] = "1"
    import ctypes as C

    import numpy as np
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd import codegen as cg
    from aesara_amd._lib import check, lib
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Plan

    if a.sum_only:
        plan = Plan.from_json(SUM_PLAN)
    else:
        plan = case_plan(next(c for c in CASES if c["name"] == "cfg2_gauss_sum"))
    f64 = torch.float64
    xs = []
    for k in range(8):
        g = torch.Generator(device="cuda")
        g.manual_seed(1 + 100 * k)
        xs.append(torch.randn(a.rows, 4096, dtype=f64, device="cuda", generator=g))
    mu = torch.tensor(0.1, dtype=f64, device="cuda")
    sigma = torch.tensor(1.3, dtype=f64, device="cuda")
    args = (lambda x: (x,)) if a.sum_only else (lambda x: (x, mu, sigma))
    ex = PlanExecutor(plan, use_graph=True, borrow=True)
    for k in range(16):
        ex(*args(xs[k % 8]))
    torch.cuda.synchronize()
    off = lib.ahip_reduce_partials_bytes() + 4096
    S = cg.TRACE_SLOTS
    rows, gaps = [], []
    for k in range(a.evals):
        # keep the stream busy in front of the traced eval (a launch after an idle stream starts
        # differently from one inside the benchmark's back-to-back sequence)
        for j in range(4):
            ex(*args(xs[(k * 5 + j) % 8]))
        ex(*args(xs[(k * 5 + 4) % 8]))
        torch.cuda.synchronize()
        raw = ex._ws[off:].view(torch.int64).cpu().numpy()
        raw = raw[:(raw.size // S) * S].reshape(-1, S)
        halves = []
        for h in range(2):                     # even / odd launch epochs: two consecutive launches
            blk = raw[h * cg.TRACE_HALF:(h + 1) * cg.TRACE_HALF]
            n = 0
            while n < blk.shape[0] and blk[n, 1] != 0:
                n += 1
            halves.append(blk[:n].copy())
        halves.sort(key=lambda b: b[:, 1].min() if b.size else 0)
        prev, st = halves
        ex._ws[off:].zero_()
        rows.append(st)
        if prev.size and st.size:
            # one clock for both launches: when did the next launch's first wavefront start,
            # counted from the previous launch's last act (its result store)
            gaps.append(((st[:, 1].min() - prev[0, 6]) * 0.01, (st[:, 1].min() - prev[:, 4].max()) * 0.01,
                         (st[0, 6] - prev[0, 6]) * 0.01))
    G = rows[0].shape[0]
    tick_us = 0.01
    out = {"workload": ("sum(x)" if a.sum_only else "exp(-(x-mu)^2/(2 sigma^2)).sum()") +
           " fp64 %dx4096, rotating over 8 inputs (MALL-cold)" % a.rows,
           "workgroups": int(G), "evals_traced": len(rows), "clock": "s_memrealtime, 100 MHz (10 ns)",
           "marks": ["entry", "first group evaluated", "stream loop done", "partial published",
                     "wg0: all partials collected", "wg0: result stored"]}
    agg = {}

    def add(name, v):
        agg.setdefault(name, []).append(v)
    for st in rows:
        t_entry = st[:, 1]
        t0 = t_entry.min()
        first = st[:, 2][st[:, 2] != 0]
        loop = st[:, 3]
        pub = st[:, 4]
        add("entry_last_wg_us", (t_entry.max() - t0) * tick_us)
        add("entry_median_us", (np.median(t_entry) - t0) * tick_us)
        if first.size:
            add("first_group_evaluated_min_us", (first.min() - t0) * tick_us)
            add("first_group_evaluated_median_us", (np.median(first) - t0) * tick_us)
            add("first_group_evaluated_max_us", (first.max() - t0) * tick_us)
        add("loop_done_first_wg_us", (loop.min() - t0) * tick_us)
        add("loop_done_median_us", (np.median(loop) - t0) * tick_us)
        add("loop_done_p90_us", (np.percentile(loop, 90) - t0) * tick_us)
        add("loop_done_last_wg_us", (loop.max() - t0) * tick_us)
        add("published_last_wg_us", (pub.max() - t0) * tick_us)
        add("publish_after_loop_median_us", float(np.median(pub - loop)) * tick_us)
        add("collected_us", (st[0, 5] - t0) * tick_us)
        add("result_stored_us", (st[0, 6] - t0) * tick_us)
        add("collect_after_last_publish_us", (st[0, 5] - pub.max()) * tick_us)
        add("store_after_collect_us", (st[0, 6] - st[0, 5]) * tick_us)
        add("wg0_loop_done_us", (st[0, 3] - t0) * tick_us)
    if gaps:
        g = np.array(gaps)
        out["consecutive_launches_us"] = {
            "result_stored(k) -> first_entry(k+1)": {"mean": round(float(g[:, 0].mean()), 3),
                                                     "min": round(float(g[:, 0].min()), 3),
                                                     "max": round(float(g[:, 0].max()), 3)},
            "last_partial_published(k) -> first_entry(k+1)": {"mean": round(float(g[:, 1].mean()), 3)},
            "result_stored(k) -> result_stored(k+1) (the period)": {"mean": round(float(g[:, 2].mean()), 3),
                                                                     "min": round(float(g[:, 2].min()), 3)}}
    out["timeline_us_from_first_entry"] = {k: {"mean": round(float(np.mean(v)), 3),
                                                 "min": round(float(np.min(v)), 3),
                                                 "max": round(float(np.max(v)), 3)}
                                           for k, v in agg.items()}
    # XCD placement of the workgroups (speed hint only)
    xcc = (rows[-1][:, 7] >> 32) & 0xf
    out["workgroups_per_xcc"] = [int((xcc == i).sum()) for i in range(8)]
    # distribution of entry / loop-done over the workgroups of the last traced eval, and how many
    # workgroups shared a physical CU (HW_ID: cu [11:8], sh [12], se [15:13])
    st = rows[-1]
    t0 = st[:, 1].min()
    ent = (st[:, 1] - t0) * tick_us
    done = (st[:, 3] - t0) * tick_us
    hist = lambda v: {("%g-%g" % (lo, hi)): int(((v >= lo) & (v < hi)).sum())       # noqa: E731
                      for lo, hi in ((0, .5), (.5, 1), (1, 2), (2, 4), (4, 8), (8, 12), (12, 16),
                                     (16, 20), (20, 24), (24, 28), (28, 40))}
    out["last_eval_entry_hist_us"] = hist(ent)
    out["last_eval_loop_done_hist_us"] = hist(done)
    hw = st[:, 7] & 0xffffffff
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    _, cnt = np.unique(cu, return_counts=True)
    out["workgroups_per_physical_cu"] = {str(int(k)): int((cnt == k).sum()) for k in np.unique(cnt)}
    out["physical_cus_used"] = int(cnt.size)
    late = ent > 2.0
    out["late_entry_workgroups"] = {"count": int(late.sum()),
                                    "block_ids_head": [int(i) for i in np.nonzero(late)[0][:24]]}
    if a.raw:
        np.savez_compressed(a.raw, stamps=np.stack(rows[-4:]))

    # event-timed per-eval time of this (traced) build and of the plain build, same rotation
    def timed(exe, iters=400):
        e0, e1 = C.c_void_p(), C.c_void_p()
        check(lib.ahip_event_create(C.byref(e0)))
        check(lib.ahip_event_create(C.byref(e1)))
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for k in range(16):
            exe(*args(xs[k % 8]))
        torch.cuda.synchronize()
        check(lib.ahip_event_record(e0, s))
        for k in range(iters):
            exe(*args(xs[k % 8]))
        check(lib.ahip_event_record(e1, s))
        torch.cuda.synchronize()
        ms = C.c_float()
        check(lib.ahip_event_elapsed_ms(e0, e1, C.byref(ms)))
        return ms.value / iters * 1e3
    out["event_us_per_eval_traced_build"] = round(timed(ex), 3)
    os.environ["AESARA_HIP_EW_TRACE"] = "0"
    ex2 = PlanExecutor(plan, use_graph=True, borrow=True)
    out["event_us_per_eval_plain_build"] = round(timed(ex2), 3)
    span = out["timeline_us_from_first_entry"]["result_stored_us"]["mean"]
    out["boundary_us"] = round(out["event_us_per_eval_traced_build"] - span, 3)
    out["note"] = ("boundary_us = event time per eval of the traced build - (first entry -> result "
                   "stored): end of one launch to the first wavefront of the next")
    print(json.dumps(out, indent=1))


SUM_PLAN = {"version": 1, "name": "sum_all_f64",
            "vars": [{"id": 0, "dtype": "float64", "shape": [None, None], "name": "x"},
                     {"id": 1, "dtype": "float64", "shape": []}],
            "inputs": [0], "outputs": [1],
            "nodes": [{"op": "CAReduce", "inputs": [0], "outputs": [1],
                       "params": {"scalar_op": "add", "axis": None, "acc_dtype": "float64"}}]}

if __name__ == "__main__":
    main()
