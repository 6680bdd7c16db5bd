"""Multi-GPU path on CPU: the sharding analysis (`aesara_amd.dist.shard_plan`), the
round structure + packed accumulator-dtype exchange (`ShardedPlan`) over a world_size-2 `gloo`
group, k logical shards in one process (`run_local_shards`), and independent-output placement.
The per-rank evaluation uses the oracle here (no GPU in this tier); tests/test_gpu_dist.py runs
the same classes over the HIP executor."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from dist_plans import (colsoftmax_plan, gemv_beta_plan, mean_plan, oracle_factory,  # noqa: E402
                        prod_plan, two_tower_plan)


def _worker(rank, world, port, case_name, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_util import CASES, case_inputs, case_plan
        from aesara_amd.dist import ShardedPlan, shard_rows

        if case_name == "colsoftmax":
            plan = colsoftmax_plan()
            ins = [np.random.default_rng(3).standard_normal((37, 5))]
            split = {0: 0}
        elif case_name == "mean":
            plan = mean_plan()
            ins = [np.random.default_rng(4).standard_normal((41, 6)).astype(np.float32)]
            split = {0: 0}
        else:
            c = next(c for c in CASES if c["name"] == case_name)
            plan, ins = case_plan(c), case_inputs(c)
            split = {0: 0, 3: 0} if case_name == "cfg5_logistic" else {0: 0}
        if case_name == "cfg4_gru_b8_f32":
            split = {0: 1, 1: 0}           # x [T, B, D] and h0 [B, H]: the batch axis
        n = ins[0].shape[split[0]]
        lo, hi = shard_rows(n, world, rank)
        local = [np.take(x, np.arange(lo, hi), axis=split[k]) if k in split else x
                 for k, x in enumerate(ins)]
        sp = ShardedPlan(plan, split, executor_factory=oracle_factory)
        outs = sp(*local)
        outs2 = sp(*local)             # second call: packed buffers are reused
        res = [np.asarray(o) for o in outs]
        for a, b in zip(res, outs2):
            np.testing.assert_array_equal(a, np.asarray(b))
        q.put((rank, sp.spec.n_exchange_rounds, [st[0] for st in sp.spec.out_state], res))
    finally:
        dist.destroy_process_group()


def _run2(case_name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("case_name", ["cfg2_gauss_sum", "cfg5_logistic"])
def test_row_sharded_allreduce_matches_unsharded(case_name):
    """BASELINE configs 2 and 5 row-sharded over 2 gloo ranks: ONE exchange round of
    accumulator-dtype (fp64) partials; logp / db within 1e-6 of the reference's output, dw within
    the reference's own fp32 error."""
    from golden_util import CASES, case_expected
    got = _run2(case_name)
    c = next(c for c in CASES if c["name"] == case_name)
    exp = case_expected(c)
    for rank, n_rounds, states, outs in got:
        assert n_rounds == 1 and all(s == "rep" for s in states)
        for k, (o, e) in enumerate(zip(outs, exp)):
            assert o.dtype == e.dtype and o.shape == e.shape
            tol = 1e-6 if e.ndim == 0 else 1e-5
            np.testing.assert_allclose(o.astype(np.float64), e.astype(np.float64), rtol=tol,
                                       atol=tol * max(1.0, float(np.abs(e).max())))
    for a, b in zip(got[0][3], got[1][3]):      # every rank holds the same combined value
        np.testing.assert_array_equal(a, b)


def test_two_exchange_rounds_and_split_output_gloo():
    """Column softmax over the split axis: max (round 0) -> sum of exp (round 1) -> row-local
    quotient: two exchange rounds, the output stays sharded."""
    got = _run2("colsoftmax")
    x = np.random.default_rng(3).standard_normal((37, 5))
    e = np.exp(x - x.max(axis=0, keepdims=True))
    want = e / e.sum(axis=0, keepdims=True)
    from aesara_amd.dist import shard_rows
    for rank, n_rounds, states, outs in got:
        assert n_rounds == 2 and states == ["split"]
        lo, hi = shard_rows(37, 2, rank)
        np.testing.assert_allclose(outs[0], want[lo:hi], rtol=1e-12)


def test_global_extent_mean_gloo():
    """mean over the split axis = Sum / Shape_i: the local extent is exchanged (summed) for the
    division while allocations would keep the local one."""
    got = _run2("mean")
    x = np.random.default_rng(4).standard_normal((41, 6)).astype(np.float32)
    for rank, n_rounds, states, outs in got:
        assert states == ["rep"]
        np.testing.assert_allclose(outs[0], x.astype(np.float64).mean(axis=0).astype(np.float32),
                                   rtol=1e-6)


def test_unprovable_plans_raise():
    from golden_util import CASES, case_plan
    from aesara_amd.dist import ShardingError, plan_split_outputs, shard_plan
    with pytest.raises(ShardingError):      # product over the split axis is not combinable here
        shard_plan(prod_plan(), {0: 0})
    with pytest.raises(ShardingError):      # beta*y would be summed world-size times
        shard_plan(gemv_beta_plan(), {1: 1, 2: 0})
    cfg5 = case_plan(next(c for c in CASES if c["name"] == "cfg5_logistic"))
    with pytest.raises(ShardingError):      # y spans the split axis but was not declared split
        shard_plan(cfg5, {0: 0})
    assert plan_split_outputs(cfg5, 0) == ["allreduce"] * 3
    scan = case_plan(next(c for c in CASES if c["name"] == "cfg4_gru_b1_f32"))
    with pytest.raises(ShardingError):      # Scan over a split sequence: replicas only
        shard_plan(scan, {0: 0})
    # replicated everywhere: a single local round, nothing exchanged
    spec = shard_plan(scan, {})
    assert spec.n_exchange_rounds == 0 and len(spec.rounds) == 1


def test_exchange_dtype_is_the_accumulator_dtype():
    from golden_util import CASES, case_plan
    from aesara_amd.dist import shard_plan
    spec = shard_plan(case_plan(next(c for c in CASES if c["name"] == "cfg5_logistic")), {0: 0, 3: 0})
    (p0, ex0), (p1, ex1) = spec.rounds
    assert [(op, dt) for _, op, dt in ex0] == [("add", "float64")] * 3 and ex1 == []
    # the float32 results are produced only after the combine
    assert {p1.vars[o].dtype for o in p1.outputs} == {"float32"}
    assert all(n.op == "Elemwise" for n in p1.nodes)


def test_logical_shards_in_one_process_match_unsharded():
    import interp
    from golden_util import CASES, case_inputs, case_plan
    from aesara_amd.dist import run_local_shards, shard_rows
    c = next(c for c in CASES if c["name"] == "cfg5_logistic")
    plan, ins = case_plan(c), case_inputs(c)
    want = interp.run_plan(plan, ins)
    for k in (1, 2, 3):
        shards = []
        for r in range(k):
            lo, hi = shard_rows(ins[0].shape[0], k, r)
            shards.append([ins[0][lo:hi], ins[1], ins[2], ins[3][lo:hi]])
        outs, spec = run_local_shards(plan, {0: 0, 3: 0}, shards, executor_factory=oracle_factory)
        for s in range(k):
            for o, w in zip(outs[s], want):
                tol = 2e-6 if w.ndim == 0 else 1e-5      # dw: the reference's own fp32 error
                np.testing.assert_allclose(np.asarray(o), w, rtol=tol,
                                           atol=tol * max(1.0, float(np.abs(w).max())))


def test_independent_outputs_are_placed_on_different_ranks():
    import interp
    from aesara_amd.dist import PlacedPlan, place_outputs
    plan = two_tower_plan()
    placement = place_outputs(plan, 2)
    # outputs 0 and 2 share the first tower (kept together), output 1 is the heavier GEMM tower
    assert sorted(map(tuple, placement)) == [(0, 2), (1,)]
    assert place_outputs(plan, 1) == [[0, 1, 2]]
    rng = np.random.default_rng(0)
    ins = [rng.standard_normal((6, 4)), rng.standard_normal((4, 4)), rng.standard_normal((4, 3))]
    want = interp.run_plan(plan, ins)
    seen = set()
    for rank in range(2):
        pp = PlacedPlan(plan, 2, rank, executor_factory=oracle_factory)
        outs = pp(*ins)
        for k, o in enumerate(outs):
            if o is not None:
                assert pp.owner(k) == rank and k not in seen
                seen.add(k)
                np.testing.assert_allclose(np.asarray(o), want[k], rtol=1e-12)
    assert seen == {0, 1, 2}
    # each rank's sub-plan holds only its own tower's nodes
    n_nodes = [len(PlacedPlan(plan, 2, r, executor_factory=oracle_factory).exec.plan.nodes)
               for r in range(2)]
    assert sum(n_nodes) == len(plan.nodes)


def test_shard_rows_partition():
    from aesara_amd.dist import shard_rows
    for n in (0, 1, 7, 4096, 16777216):
        for w in (1, 2, 3, 8):
            blocks = [shard_rows(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1


def test_random_plans_shard_soundly():
    """Soundness of the sharding analysis on random multi-node plans (the generator of the GPU
    DAG fuzz: Elemwise, transposes, row slices, IncSubtensor, gathers / scatters, axis sums, row
    chains, Dot22 on integer-valued float64 — exact arithmetic): with the (R, C) inputs split by
    rows and the (C, R) input by columns, ``shard_plan`` either refuses (ShardingError) or every
    logical shard's result equals the unsharded value bit for bit (its own block where the output
    stays split)."""
    import interp
    from aesara_amd.dist import ShardingError, run_local_shards, shard_rows
    from test_gpu_fuzz import _rand_dag
    accepted = refused = 0
    for seed in range(6):
        rng = np.random.default_rng(9000 + seed)
        for trial in range(40):
            R, C = (int(v) for v in rng.choice([4, 6, 7, 8, 9], 2, replace=False))
            plan, shapes, idx_in = _rand_dag(rng, R, C, int(rng.integers(2, 10)))
            args = [rng.integers(-R, R, 5).astype("int64") if v == idx_in
                    else rng.integers(-3, 4, shapes[v]).astype("float64") for v in plan.inputs]
            want = interp.run_plan(plan, args)
            k = int(rng.integers(2, 4))
            split = {0: 0, 1: 0, 2: 1}
            shards = []
            for r in range(k):
                lo, hi = shard_rows(R, k, r)
                shards.append([args[0][lo:hi], args[1][lo:hi], args[2][:, lo:hi], args[3]])
            try:
                outs, spec = run_local_shards(plan, split, shards, executor_factory=oracle_factory)
            except ShardingError:
                refused += 1
                continue
            accepted += 1
            for r in range(k):
                lo, hi = shard_rows(R, k, r)
                for o, w, st in zip(outs[r], want, spec.out_state):
                    o = np.asarray(o)
                    if st[0] == "split":
                        sl = [slice(None)] * w.ndim
                        sl[st[1]] = slice(lo, hi)
                        w = w[tuple(sl)]
                    assert o.shape == w.shape and np.array_equal(o, w), (seed, trial, r, st, plan.pretty())
    assert accepted >= 40 and refused >= 20, (accepted, refused)


def test_scan_is_sharded_along_the_batch_axis():
    """BASELINE config 4 with a matrix state: sequences ``[T, B, D]`` and the initial state
    ``[B, H]`` split along the batch axis — the step is row-local (checked by the same analysis on
    the inner plan), so every shard runs the loop on its rows with NO exchange and the outputs stay
    split on the batch axis; bit-equal to the unsharded blocks.  Gradient Scans (mit-mot) and a
    split along time are refused."""
    import interp
    from golden_util import CASES, case_inputs, case_plan
    from aesara_amd.dist import ShardingError, run_local_shards, shard_plan, shard_rows
    c = next(c for c in CASES if c["name"] == "cfg4_gru_b8_f32")
    plan, ins = case_plan(c), case_inputs(c)
    want = interp.run_plan(plan, ins)
    B = ins[0].shape[1]
    for k in (2, 3):
        shards = []
        for r in range(k):
            lo, hi = shard_rows(B, k, r)
            shards.append([ins[0][:, lo:hi], ins[1][lo:hi]] + list(ins[2:]))
        outs, spec = run_local_shards(plan, {0: 1, 1: 0}, shards, executor_factory=oracle_factory)
        assert spec.n_exchange_rounds == 0 and len(spec.rounds) == 1
        assert spec.out_state == [("split", 1), ("split", 0)]
        for r in range(k):
            lo, hi = shard_rows(B, k, r)
            assert np.array_equal(np.asarray(outs[r][0]), want[0][:, lo:hi])
            assert np.array_equal(np.asarray(outs[r][1]), want[1][lo:hi])
    with pytest.raises(ShardingError):          # the time axis cannot be split
        shard_plan(plan, {0: 0})
    with pytest.raises(ShardingError):          # h0 replicated next to a split x: states disagree
        shard_plan(plan, {0: 1})
    bptt = case_plan(next(c for c in CASES if c["name"] == "gru_bptt_b4_f32"))
    with pytest.raises(ShardingError):          # gradient Scan / Reshape of the split arrays
        shard_plan(bptt, {0: 1, 1: 0})


def test_batch_sharded_scan_over_two_gloo_ranks():
    """``ShardedPlan`` over a world-size-2 gloo group on BASELINE config 4 (matrix state) split
    along the batch: no exchange round, every rank returns its rows of all hidden states."""
    from golden_util import CASES, case_expected
    got = _run2("cfg4_gru_b8_f32")
    c = next(c for c in CASES if c["name"] == "cfg4_gru_b8_f32")
    hs, hT = case_expected(c)
    B = hs.shape[1]
    for rank, n_rounds, states, outs in got:
        assert n_rounds == 0 and states == ["split", "split"]
        lo, hi = (0, B // 2) if rank == 0 else (B // 2, B)
        np.testing.assert_allclose(outs[0], hs[:, lo:hi], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(outs[1], hT[lo:hi], rtol=1e-5, atol=1e-6)
