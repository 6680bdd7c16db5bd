"""RCCL on the GPU box: the collective path runs on hardware before an 8-GPU box
does it for the first time — through the C-ABI communicator (``ahip_comm_*`` / ``ahip_allreduce``:
RCCL enqueued by the shim on the launch stream, also as a launch-list entry) and through
``torch.distributed``'s "nccl" backend (= RCCL), both with a world of ONE rank (an identity that
still goes through RCCL's all-reduce kernels)."""
import ctypes as C
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg5(n):
    from golden_util import CASES, case_inputs, case_plan
    c = next(c for c in CASES if c["name"] == "cfg5_logistic")
    plan = case_plan(c)
    rng = np.random.default_rng(6)
    X = rng.standard_normal((n, 256)).astype("float32")
    w = (np.random.default_rng(7).standard_normal(256) / 16).astype("float32")
    y = (np.random.default_rng(8).random(n) < 0.5).astype("float32")
    return plan, [X, w, np.asarray(0.1, "float32"), y]


@pytest.fixture(scope="module")
def comm():
    from aesara_amd.dist import HipComm
    c = HipComm(world=1, rank=0)
    yield c
    c.close()


def test_abi_allreduce_world1_is_identity_and_records_into_a_launch_list(comm):
    import torch
    from aesara_amd._lib import check, lib
    for dt in (torch.float64, torch.float32, torch.int64, torch.int32):
        x = (torch.arange(1000, device="cuda") * 3 - 7).to(dt)
        want = x.clone()
        comm.all_reduce(x, "add")
        comm.all_reduce(x, "maximum")
        torch.cuda.synchronize()
        assert torch.equal(x, want)
    assert lib.ahip_comm_size(comm._h) == 1 and lib.ahip_comm_rank(comm._h) == 0
    # as a launch-list entry between two kernels: fill -> all-reduce -> copy, replayed 3 times
    a = torch.zeros(4096, dtype=torch.float64, device="cuda")
    b = torch.zeros(4096, dtype=torch.float64, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    val = C.c_double(2.5)
    lst = C.c_void_p()
    check(lib.ahip_list_begin())
    try:
        check(lib.ahip_fill(10, C.byref(val), C.c_void_p(a.data_ptr()), 4096, stream))
        comm.all_reduce(a, "add")                          # recorded, not executed
        sh = (C.c_int64 * 1)(4096)
        st = (C.c_int64 * 1)(1)
        check(lib.ahip_copy_strided(10, 1, sh, C.c_void_p(a.data_ptr()), st, C.c_void_p(b.data_ptr()), st,
                                    0, stream))
    finally:
        check(lib.ahip_list_end(C.byref(lst)))
    torch.cuda.synchronize()
    assert float(a.sum()) == 0.0 and lib.ahip_list_length(lst) == 3      # nothing ran while recording
    for _ in range(3):
        b.zero_()
        check(lib.ahip_list_run(lst, stream))
        torch.cuda.synchronize()
        assert float(b.min()) == 2.5 and float(b.max()) == 2.5
    lib.ahip_list_destroy(lst)


@pytest.mark.parametrize("use_graph", [False, True])
def test_sharded_cfg5_and_two_round_softmax_over_the_abi_communicator(comm, use_graph):
    import interp
    import torch
    from dist_plans import colsoftmax_plan
    from aesara_amd.dist import ShardedPlan
    plan, ins = _cfg5(16384)
    want = interp.run_plan(plan, ins)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if a.ndim else torch.tensor(a.item(), device="cuda")
           for a in ins]
    sp = ShardedPlan(plan, {0: 0, 3: 0}, group=comm, use_graph=use_graph, force_collectives=True)
    for _ in range(3):
        outs = sp(*dev)
        for o, w in zip(outs, want):
            tol = 1e-6 if w.ndim == 0 else 1e-5
            np.testing.assert_allclose(o.cpu().numpy().astype(np.float64), w.astype(np.float64),
                                       rtol=tol, atol=tol * max(1.0, float(np.abs(w).max())))
    x = np.random.default_rng(3).standard_normal((1037, 65))
    e = np.exp(x - x.max(axis=0, keepdims=True))
    sp2 = ShardedPlan(colsoftmax_plan(), {0: 0}, group=comm, use_graph=use_graph, force_collectives=True)
    assert sp2.spec.n_exchange_rounds == 2
    xd = torch.from_numpy(x).cuda()
    for _ in range(4):
        (o,) = sp2(xd)
        np.testing.assert_allclose(o.cpu().numpy(), e / e.sum(axis=0, keepdims=True), rtol=1e-12)
    if use_graph:
        # ONE launch list per sharded evaluation: the first evaluation of a signature builds the
        # packed buffers, the second records rounds + all-reduces into a single list, the rest
        # replay it (no Python between the rounds)
        assert sp.single_list and sp.replays >= 1, sp.replays
        assert sp2.single_list and sp2.replays >= 2, sp2.replays
        (lst, _final, _keep, _exts) = next(iter(sp2._lists.values()))
        from aesara_amd._lib import lib
        assert lib.ahip_list_length(lst) >= 4        # >= 2 kernels + 2 all-reduces in ONE list
        assert comm.recorded >= 2


def test_single_list_signatures_keep_their_arenas_and_are_evicted(comm):
    """Two signatures replayed ALTERNATELY from their own recorded lists (another batch shape, a
    second input buffer, a transposed view of the same buffer): every entry owns the arenas its
    list addresses (the executors' one slot is overwritten by the next recording), allocations
    made in between must not be written by a replay, and the cache is LRU-bounded with
    ``ahip_list_destroy`` + the communicator's ``recorded`` count following."""
    import torch
    from dist_plans import colsoftmax_plan
    from aesara_amd.dist import ShardedPlan

    def want(x):
        e = np.exp(x - x.max(axis=0, keepdims=True))
        return e / e.sum(axis=0, keepdims=True)
    rng = np.random.default_rng(11)
    sp = ShardedPlan(colsoftmax_plan(), {0: 0}, group=comm, use_graph=True, force_collectives=True)
    base = comm.recorded
    xa, xb = rng.standard_normal((1037, 65)), rng.standard_normal((523, 65))
    xs = rng.standard_normal((96, 96))
    da, db, ds = (torch.from_numpy(v).cuda() for v in (xa, xb, xs))
    for _ in range(3):                       # a: ordinary, record, replay / b: the same
        for d, x in ((da, xa), (db, xb)):
            (o,) = sp(d)
            np.testing.assert_allclose(o.cpu().numpy(), want(x), rtol=1e-12)
    assert len(sp._lists) == 2 and comm.recorded == base + 2
    guard = []
    for it in range(6):
        # memory the allocator hands out now is where a freed arena would have been
        guard.append(torch.full((1 << 16,), float(it), dtype=torch.float64, device="cuda"))
        for d, x in ((da, xa), (db, xb)):
            (o,) = sp(d)
            np.testing.assert_allclose(o.cpu().numpy(), want(x), rtol=1e-12)
        for i, g in enumerate(guard):
            assert float(g.min()) == float(i) == float(g.max())
    # same buffer, same shape, other strides: another signature, not the cached launches
    for _ in range(3):
        (o,) = sp(ds)
        np.testing.assert_allclose(o.cpu().numpy(), want(xs), rtol=1e-12)
        (o,) = sp(ds.t())
        np.testing.assert_allclose(o.cpu().numpy(), want(xs.T), rtol=1e-12)
    n = len(sp._lists)
    assert 3 <= n <= sp.LISTS_MAX
    sp.LISTS_MAX = 2
    fresh = [torch.from_numpy(xb).cuda() for _ in range(4)]      # fresh batch tensors: new signatures
    for d in fresh:
        for _ in range(3):
            (o,) = sp(d)
            np.testing.assert_allclose(o.cpu().numpy(), want(xb), rtol=1e-12)
    assert len(sp._lists) <= 2 and comm.recorded == base + len(sp._lists)
    sp.close()
    assert not sp._lists and comm.recorded == base


def test_torch_distributed_nccl_backend_world1():
    """``init_process_group("nccl")`` (RCCL) with one rank: the sharded config-5 plan, the packed
    fp64 all-reduce, and the bucketed ASYNC all-reduce of config 2's partials (bench.py's N > 1
    path: ring slots written by ``out=``, one collective per bucket, handles waited later)."""
    import interp
    import torch
    import torch.distributed as dist
    from golden_util import CASES, case_plan
    from aesara_amd.dist import ShardedFunction, ShardedPlan, plan_split_outputs
    from aesara_amd.executor import PlanExecutor
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        plan, ins = _cfg5(8192)
        want = interp.run_plan(plan, ins)
        dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if a.ndim else torch.tensor(a.item(), device="cuda")
               for a in ins]
        sp = ShardedPlan(plan, {0: 0, 3: 0}, group=dist.group.WORLD, use_graph=True, force_collectives=True)
        for _ in range(3):
            outs = sp(*dev)
            for o, w in zip(outs, want):
                tol = 1e-6 if w.ndim == 0 else 1e-5
                np.testing.assert_allclose(o.cpu().numpy().astype(np.float64), w.astype(np.float64),
                                           rtol=tol, atol=tol * max(1.0, float(np.abs(w).max())))
        # bucketed async path of bench.py (world 1: the reducer is called directly)
        c2 = case_plan(next(c for c in CASES if c["name"] == "cfg2_gauss_sum"))
        ex = PlanExecutor(c2, use_graph=True, borrow=True)
        kinds = plan_split_outputs(c2, 0)
        assert kinds == ["allreduce"]
        x = torch.randn(512, 384, dtype=torch.float64, device="cuda")
        mu = torch.tensor(0.1, dtype=torch.float64, device="cuda")
        sg = torch.tensor(1.3, dtype=torch.float64, device="cuda")
        ring = torch.zeros(16, dtype=torch.float64, device="cuda")
        handles = []
        for i in range(16):
            ex(x, mu, sg, out=[ring[i]])
            if (i + 1) % 8 == 0:
                h = dist.all_reduce(ring[i - 7:i + 1], op=dist.ReduceOp.SUM, async_op=True)
                handles.append(h)
        for h in handles:
            h.wait()
        torch.cuda.synchronize()
        ref = torch.exp(-(x - 0.1) ** 2 / (2 * 1.3 ** 2)).sum().item()
        np.testing.assert_allclose(ring.cpu().numpy(), np.full(16, ref), rtol=1e-12)
        red = ShardedFunction(lambda b: [b], kinds, group=dist.group.WORLD)
        (o,), hs = red(ring[:8], async_op=True)
        assert hs == [] and o.data_ptr() == ring.data_ptr()     # world 1: nothing to combine
        _bench_transport_guard(dist, torch)
    finally:
        dist.destroy_process_group()


def _bench_transport_guard(dist, torch):
    """``bench.guard_first_exchange`` (round 6): the first sharded evaluation of the C-ABI transport
    under a wall-clock guard — an event polled from the host, ``ahip_comm_abort`` and the fall-back
    decision over ``torch.distributed`` on a time-out or an exception.  (An N > 1 run is what it is
    for; a box with one GPU exercises every line of it except the collective's peers.)"""
    import importlib.util
    import os
    from aesara_amd._lib import check, lib
    from aesara_amd.dist import HipComm
    spec = importlib.util.spec_from_file_location(
        "bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    buf = torch.ones(64, dtype=torch.float64, device="cuda")
    # (1) a healthy exchange: completes, the communicator stays
    g = HipComm(world=1, rank=0)
    assert bench.guard_first_exchange(lambda: g.all_reduce(buf), g, C, lib, check, torch, dist, 0) is None
    assert g._h
    assert bench.wait_for_stream(C, lib, check, torch, 5.0) is True
    # (2) the exchange raises: the communicator is aborted, the caller is told to fall back
    def boom():
        raise RuntimeError("no peer")
    err = bench.guard_first_exchange(boom, g, C, lib, check, torch, dist, 1)
    assert err and "no peer" in err and not g._h
    # (3) the guard's time limit (0 s: "did not complete in time" without waiting): abort + fall-back
    g2 = HipComm(world=1, rank=0)
    old = bench.TRANSPORT_GUARD_S
    bench.TRANSPORT_GUARD_S = 0.0
    try:
        err = bench.guard_first_exchange(lambda: g2.all_reduce(buf), g2, C, lib, check, torch, dist, 1)
    finally:
        bench.TRANSPORT_GUARD_S = old
    assert err and "did not complete" in err and not g2._h
    torch.cuda.synchronize()
    assert float(buf.sum().item()) == 64.0
