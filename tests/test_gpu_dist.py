"""Sharded execution over the HIP executor on ONE device: k logical shards in one
process, a 2-process gloo group with both ranks on cuda:0, replayed rounds writing straight into
the packed exchange buffer, and independent-output placement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

pytestmark = pytest.mark.gpu


def _cfg5(n=None):
    from golden_util import CASES, case_inputs, case_plan
    c = next(c for c in CASES if c["name"] == "cfg5_logistic")
    plan, ins = case_plan(c), case_inputs(c)
    if n is not None:
        rng = np.random.default_rng(11)
        D = ins[0].shape[1]
        ins = [rng.standard_normal((n, D)).astype(np.float32), ins[1], ins[2],
               (rng.random(n) < 0.5).astype(np.float32)]
    return plan, ins


@pytest.mark.parametrize("k", [2, 3])
@pytest.mark.parametrize("n", [None, 4099])
def test_logical_shards_hip_match_oracle(k, n):
    import interp
    import torch
    from aesara_amd.dist import run_local_shards, shard_rows
    plan, ins = _cfg5(n)
    want = interp.run_plan(plan, ins)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ins]
    shards = []
    for r in range(k):
        lo, hi = shard_rows(ins[0].shape[0], k, r)
        shards.append([dev[0][lo:hi], dev[1], dev[2], dev[3][lo:hi]])
    outs, spec = run_local_shards(plan, {0: 0, 3: 0}, shards)
    assert spec.n_exchange_rounds == 1
    for s in range(k):
        for o, w in zip(outs[s], want):
            o = o.cpu().numpy()
            assert o.dtype == w.dtype and o.shape == w.shape
            tol = 1e-6 if w.ndim == 0 else 1e-5
            np.testing.assert_allclose(o.astype(np.float64), w.astype(np.float64), rtol=tol,
                                       atol=tol * max(1.0, float(np.abs(w).max())))


def test_sharded_plan_world1_replay_writes_into_packed_buffer():
    """world_size 1 (no collective issued): the replayed round-0 kernels must write their fp64
    partials into the persistent packed buffer (pointer-stable), round 1 casts them."""
    import interp
    import torch
    from aesara_amd.dist import ShardedPlan
    plan, ins = _cfg5(8192)
    want = interp.run_plan(plan, ins)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ins]
    sp = ShardedPlan(plan, {0: 0, 3: 0}, use_graph=True)
    for it in range(4):
        outs = sp(*dev)
        (bufs, views, targets) = next(iter(sp._packs.values()))
        assert len(bufs) == 1 and next(iter(bufs.values())).dtype == torch.float64
        assert next(iter(bufs.values())).numel() == 2 + ins[0].shape[1]
        for o, w in zip(outs, want):
            tol = 1e-6 if w.ndim == 0 else 1e-5
            np.testing.assert_allclose(o.cpu().numpy().astype(np.float64), w.astype(np.float64),
                                       rtol=tol, atol=tol * max(1.0, float(np.abs(w).max())))
        # the partial logp in the packed buffer is the fp64 accumulator, not a rounded fp32
        packed = next(iter(bufs.values())).cpu().numpy()
        assert abs(packed[0] - float(want[0])) <= 1e-6 * abs(float(want[0]))


def test_two_round_column_softmax_hip():
    import torch
    from dist_plans import colsoftmax_plan
    from aesara_amd.dist import run_local_shards, shard_rows
    x = np.random.default_rng(3).standard_normal((1037, 65))
    e = np.exp(x - x.max(axis=0, keepdims=True))
    want = e / e.sum(axis=0, keepdims=True)
    xd = torch.from_numpy(x).cuda()
    shards = [[xd[slice(*shard_rows(1037, 3, r))]] for r in range(3)]
    outs, spec = run_local_shards(colsoftmax_plan(), {0: 0}, shards)
    assert spec.n_exchange_rounds == 2
    for r in range(3):
        lo, hi = shard_rows(1037, 3, r)
        np.testing.assert_allclose(outs[r][0].cpu().numpy(), want[lo:hi], rtol=1e-12)


def test_placed_outputs_hip():
    import interp
    import torch
    from dist_plans import two_tower_plan
    from aesara_amd.dist import PlacedPlan
    plan = two_tower_plan()
    rng = np.random.default_rng(0)
    ins = [rng.standard_normal((600, 40)), rng.standard_normal((48, 40)), rng.standard_normal((40, 33))]
    want = interp.run_plan(plan, ins)
    dev = [torch.from_numpy(a).cuda() for a in ins]
    seen = set()
    for rank in range(2):
        outs = PlacedPlan(plan, 2, rank)(*dev)
        for k, o in enumerate(outs):
            if o is not None:
                seen.add(k)
                np.testing.assert_allclose(o.cpu().numpy(), want[k], rtol=1e-12)
    assert seen == {0, 1, 2}


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aesara_amd.dist import ShardedPlan, shard_rows
        plan, ins = _cfg5(6000)
        lo, hi = shard_rows(6000, world, rank)
        dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in
               (ins[0][lo:hi], ins[1], ins[2], ins[3][lo:hi])]
        sp = ShardedPlan(plan, {0: 0, 3: 0}, use_graph=True)
        for _ in range(3):
            outs = sp(*dev)
        torch.cuda.synchronize()
        q.put((rank, [o.cpu().numpy() for o in outs]))
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_over_hip_executor():
    import interp
    import torch.multiprocessing as mp
    plan, ins = _cfg5(6000)
    want = interp.run_plan(plan, ins)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, outs in got:
        for o, w in zip(outs, want):
            tol = 1e-6 if w.ndim == 0 else 1e-5
            np.testing.assert_allclose(o.astype(np.float64), w.astype(np.float64), rtol=tol,
                                       atol=tol * max(1.0, float(np.abs(w).max())))
    for a, b in zip(got[0][1], got[1][1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("use_graph", [False, True])
def test_random_plans_shard_soundly_on_the_hip_executor(use_graph):
    """The sharding soundness fuzz of tests/test_dist_gloo.py with the real executor per logical
    shard (eager and replayed rounds): every accepted random plan is bit-equal to the unsharded
    oracle value on every shard (its block where the output stays split), two calls each."""
    import interp
    import torch
    from aesara_amd.dist import ShardingError, run_local_shards, shard_rows
    from test_gpu_fuzz import _rand_dag
    accepted = 0
    rng = np.random.default_rng(9100)
    for trial in range(120):
        R, C = (int(v) for v in rng.choice([4, 6, 7, 8, 9], 2, replace=False))
        plan, shapes, idx_in = _rand_dag(rng, R, C, int(rng.integers(2, 10)))
        k = int(rng.integers(2, 4))
        split = {0: 0, 1: 0, 2: 1}
        for call in range(2):
            args = [rng.integers(-R, R, 5).astype("int64") if v == idx_in
                    else rng.integers(-3, 4, shapes[v]).astype("float64") for v in plan.inputs]
            want = interp.run_plan(plan, args)
            dev = [torch.from_numpy(a).cuda() for a in args]
            shards = []
            for r in range(k):
                lo, hi = shard_rows(R, k, r)
                shards.append([dev[0][lo:hi], dev[1][lo:hi], dev[2][:, lo:hi], dev[3]])
            try:
                outs, spec = run_local_shards(plan, split, shards, use_graph=use_graph)
            except ShardingError:
                break
            accepted += call == 0
            for r in range(k):
                lo, hi = shard_rows(R, k, r)
                for o, w, st in zip(outs[r], want, spec.out_state):
                    o = o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o)
                    if st[0] == "split":
                        sl = [slice(None)] * w.ndim
                        sl[st[1]] = slice(lo, hi)
                        w = w[tuple(sl)]
                    assert o.shape == w.shape and np.array_equal(o, w), (trial, call, r, st, plan.pretty())
    assert accepted >= 15, accepted


@pytest.mark.parametrize("k", [2, 3])
def test_batch_sharded_scan_on_the_hip_executor(k):
    """BASELINE config 4 (matrix state) split along the batch into k logical shards, each running
    the persistent kernel on its rows (no exchange round): every shard's hidden states equal the
    unsharded run's rows (same kernel arithmetic per row block up to the GEMM tiling of the hoisted
    products)."""
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd.dist import run_local_shards, shard_rows
    from aesara_amd.executor import PlanExecutor
    plan = case_plan(next(c for c in CASES if c["name"] == "cfg4_gru_b8_f32"))
    T, B, H = 24, 40, 128
    g = torch.Generator(device="cuda")
    g.manual_seed(31)
    x = torch.randn(T, B, H, device="cuda", generator=g) * 0.1
    h0 = torch.randn(B, H, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    whole = PlanExecutor(plan)(x, h0, *Ws)
    shards = []
    for r in range(k):
        lo, hi = shard_rows(B, k, r)
        shards.append([x[:, lo:hi].contiguous(), h0[lo:hi].contiguous()] + Ws)
    outs, spec = run_local_shards(plan, {0: 1, 1: 0}, shards)
    assert spec.n_exchange_rounds == 0 and spec.out_state == [("split", 1), ("split", 0)]
    for r in range(k):
        lo, hi = shard_rows(B, k, r)
        assert torch.allclose(outs[r][0], whole[0][:, lo:hi], rtol=1e-4, atol=2e-6)
        assert torch.allclose(outs[r][1], whole[1][lo:hi], rtol=1e-4, atol=2e-6)
