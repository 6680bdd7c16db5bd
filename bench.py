#!/usr/bin/env python
"""Benchmark of the HIP linker hot path (driver contract: see the repo prompt / DESIGN.md §5).

Workload at N=1 = BASELINE.json configs[1]: the fused Elemwise chain
``exp(-(x-mu)**2 / (2*sigma**2)).sum()`` on a float64 4096x4096 matrix — the plan is the one the
HIP linker lowers from the reference's FAST_RUN graph (tests/golden: cfg2_gauss_sum; shapes are
dynamic, so the same plan runs at 4096x4096).  One "step" = one evaluation of the compiled
function: one launch-list replay (a single host call into libaesara_hip.so) that issues the ONE
fused Elemwise+Sum kernel (in-kernel deterministic finalize), with ``x`` resident in HBM.

N>1 (weak scaling): every rank evaluates its own 4096x4096 row block of a (N*4096)x4096 matrix;
the CAReduce partial is summed over ranks with one RCCL all-reduce per eval (issued
asynchronously so consecutive evals pipeline) — the only collective, SURVEY §8e.

Prints ONE JSON line (rank 0).  ``roofline`` is measured live with HIP events on the launch
stream; ``cpu_baseline`` times the oracle's C port of the reference C linker's loops on the
host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guide: 8.0 TB/s; ~6.3 TB/s achievable)
ROWS, COLS = 4096, 4096
ALGO_BYTES = ROWS * COLS * 8   # x read once (134 217 728 B per eval)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph (per-node launches)")
    args = ap.parse_args()

    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # test hooks (single-GPU boxes): AESARA_BENCH_BACKEND=gloo + AESARA_BENCH_ONE_DEVICE=1 run
    # the N>1 control flow (ring slots, bucketed async all-reduce) with every rank on cuda:0
    backend = os.environ.get("AESARA_BENCH_BACKEND", "nccl")
    if os.environ.get("AESARA_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from golden_util import CASES, case_plan
    from aesara_amd._lib import check, lib
    from aesara_amd.executor import PlanExecutor

    plan = case_plan(next(c for c in CASES if c["name"] == "cfg2_gauss_sum"))
    ex = PlanExecutor(plan, use_graph=not args.eager)

    # synthetic input of the named shape, generated on device (rank-specific row block)
    g = torch.Generator(device="cuda")
    g.manual_seed(1 + rank)
    x = torch.randn(ROWS, COLS, dtype=torch.float64, device="cuda", generator=g)
    mu = torch.tensor(0.1, dtype=torch.float64, device="cuda")
    sigma = torch.tensor(1.3, dtype=torch.float64, device="cuda")

    from aesara_amd.dist import ShardedFunction, plan_split_outputs

    kinds = plan_split_outputs(plan, 0)           # ["allreduce"]: Sum over the split row axis
    # ring of result slots: the fused kernel of eval i writes its partial straight into
    # ring[i % R] (executor out=), and every BUCKET evals ONE asynchronous RCCL all-reduce sums
    # a bucket of partials over the ranks (bucketed: the 8-byte payload is latency-bound on xGMI),
    # so consecutive evals pipeline behind the collective
    R, BUCKET = 32, 8
    ring = torch.zeros(R, dtype=torch.float64, device="cuda")
    slots = [ring[i] for i in range(R)]
    state = {"i": 0}
    reducer = ShardedFunction(lambda bucket: [bucket], kinds)

    def step():
        i = state["i"]
        state["i"] = i + 1
        if world == 1:
            ex(x, mu, sigma)
            return None
        ex(x, mu, sigma, out=[slots[i % R]])
        if (i + 1) % BUCKET == 0:
            lo = (i + 1 - BUCKET) % R
            _, hs = reducer(ring[lo:lo + BUCKET], async_op=True)
            return hs[0]
        return None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # correctness of the benchmarked path against an fp64 restatement on the same data
    (out,) = ex(x, mu, sigma)
    want = torch.exp(-(x - 0.1) ** 2 / (2 * 1.3 ** 2)).sum()
    rel = abs(out.item() - want.item()) / abs(want.item())
    assert rel < 1e-9, f"benchmark result mismatch: rel err {rel}"

    h = None
    for _ in range(args.warmup):
        h = step()
    if h is not None:
        h.wait()
    barrier()

    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    check(lib.ahip_event_create(C.byref(ev0)))
    check(lib.ahip_event_create(C.byref(ev1)))

    barrier()
    t0 = time.perf_counter()
    check(lib.ahip_event_record(ev0, stream))
    handles = []
    for _ in range(args.steps):
        h = step()
        if h is not None:
            handles.append(h)
            if len(handles) > 2:   # <= 3 buckets (24 evals) in flight: ring slots never reused early
                handles.pop(0).wait()
    check(lib.ahip_event_record(ev1, stream))
    for h in handles:
        h.wait()
    barrier()
    t1 = time.perf_counter()
    ms = C.c_float()
    check(lib.ahip_event_elapsed_ms(ev0, ev1, C.byref(ms)))

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = elapsed.item()

    if rank == 0:
        dev_ms_per_eval = ms.value / args.steps          # device time per eval on the stream
        achieved = ALGO_BYTES / (dev_ms_per_eval * 1e-3) / 1e9
        res = {
            "metric": "compiled-fn evals/sec + achieved HBM GB/s (elemwise) / MFMA % (gemm), 1->8 GPU",
            "value": args.steps * world / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: fused Elemwise exp(-(x-mu)^2/2sigma^2).sum(), "
                                   "fp64 4096x4096 per GPU, inputs resident in HBM, launch-list replay",
                       "rows_per_gpu": ROWS, "cols": COLS,
                       "parallelism": "row-sharded x%d, bucketed async RCCL all-reduce of the CAReduce "
                                      "partials (8 evals per collective)" % world
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(),
                         "kernel_ms": dev_ms_per_eval, "algorithmic_bytes": ALGO_BYTES},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(np)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_bench_traffic.json: FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md §HBM, + WRITE_SIZE); null when no profile has been committed."""
    path = os.path.join(ROOT, "profiles", "r01_bench_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(np):
    """Oracle C port of the reference C linker's loops for the same graph, same shape, on the
    host: bounded sample (a few evals, ~0.2 s each), single thread like the reference default."""
    import cport
    xh = np.random.default_rng(1).standard_normal((ROWS, COLS))
    cport.cfg2_eval(xh, 0.1, 1.3)  # warm-up (page faults)
    n, t0 = 0, time.perf_counter()
    while n < 20 and time.perf_counter() - t0 < 10.0:
        cport.cfg2_eval(xh, 0.1, 1.3)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "%d evals of the same 4096x4096 fp64 graph (oracle/c_port.c: unfused "
                      "Composite loop + Sum loop as the reference C linker runs them)" % n,
            "ms_per_eval": dt * 1e3, "host_cores_visible": os.cpu_count()}


if __name__ == "__main__":
    main()
