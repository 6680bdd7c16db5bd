"""Buffer ownership on the GPU: in-place updates never corrupt live views;
replayed calls hand out fresh outputs unless borrowed and still raise on a bad index (medium)."""
import numpy as np
import pytest

from test_host_logic import _alias_plan

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("view_is_output", [True, False])
@pytest.mark.parametrize("use_graph", [False, True])
def test_view_of_updated_alloc_keeps_its_values(view_is_output, use_graph):
    from aesara_amd.executor import PlanExecutor
    ex = PlanExecutor(_alias_plan(view_is_output), use_graph=use_graph)
    y = np.array([7.0, 8.0, 9.0])
    for _ in range(3):
        a, r = [o.cpu().numpy() for o in ex(np.float64(1.5), np.int64(4), y)]
        z = np.full((4, 3), 1.5)
        want_r = z.copy()
        want_r[0] = y
        np.testing.assert_array_equal(r, want_r)
        np.testing.assert_array_equal(a, z.T if view_is_output else z.T.sum(axis=0))


def test_replay_outputs_are_fresh_unless_borrowed():
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd.executor import PlanExecutor
    plan = case_plan(next(c for c in CASES if c["name"] == "cfg1b_matrix_add"))
    x = torch.ones(64, 64, dtype=torch.float64, device="cuda")
    for borrow in (False, True):
        ex = PlanExecutor(plan, use_graph=True, borrow=borrow)
        outs = []
        for k in range(4):
            y = torch.full((64, 64), float(k), dtype=torch.float64, device="cuda")
            outs.append(ex(x, y)[0])
        torch.cuda.synchronize()
        if borrow:      # function-owned buffer: later calls overwrite what was handed out
            assert len({o.data_ptr() for o in outs[1:]}) < 3
        else:           # every call's result survives the following calls
            for k, o in enumerate(outs):
                assert float(o[0, 0]) == 1.0 + k, (k, float(o[0, 0]))
            # ... and without a device copy: the recorded launch is re-pointed at a newly
            # allocated tensor (the output is a rebindable range of the launch list)
            ents = list(ex._graphs.values())
            assert ents and all(e[8] for e in ents), [e[8] for e in ents]
            arena_ptr = ents[0][2][0].data_ptr()
            assert all(o.data_ptr() != arena_ptr for o in outs[1:])


def test_replayed_call_with_bad_index_raises():
    import torch
    from golden_util import CASES, case_inputs, case_plan
    from aesara_amd.executor import PlanExecutor
    c = next(c for c in CASES if c["name"] == "advsub1")
    plan, ins = case_plan(c), case_inputs(c)
    ipos = next(k for k, a in enumerate(ins) if np.asarray(a).dtype.kind == "i" and np.asarray(a).ndim == 1)
    ex = PlanExecutor(plan, use_graph=True)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() if np.asarray(a).ndim else a for a in ins]
    for _ in range(3):
        ex(*dev)                                   # recorded, then replayed
    bad = dev[ipos].clone()
    bad[0] = 10 ** 6
    good = dev[ipos].clone()
    dev[ipos].copy_(bad)                            # same tensor, new values: the replay fast path
    ex(*dev)                                        # the check is deferred (no pipeline stall) ...
    with pytest.raises(IndexError):
        ex.check()                                  # ... and reported once the flag has landed
    dev[ipos].copy_(good)
    ex(*dev)
    ex.check()                                      # flag was reset
    dev[ipos].copy_(bad)
    ex(*dev)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):                 # without check(): the next calls report it
        for _ in range(3):
            ex(*dev)
            torch.cuda.synchronize()


def test_replay_arena_is_packed_by_lifetimes():
    """Static buffer plan: the replay arena of a many-intermediate plan is ONE
    buffer whose size is the peak of the live set, smaller than the sum of all allocations, and
    replays compute the same values in it call after call."""
    from golden_util import CASES, assert_matches, case_expected, case_inputs, case_plan
    from aesara_amd.executor import PlanExecutor
    for name in ("nll_classifier_float32", "lstm_bptt_float32", "mlp_layers_float32"):
        c = next(c for c in CASES if c["name"] == name)
        ex = PlanExecutor(case_plan(c), use_graph=True)
        ins = case_inputs(c)
        for _ in range(4):
            got = [o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o) for o in ex(*ins)]
            assert_matches(c, got, case_expected(c), "packed arena")
        total, naive = ex.arena_bytes
        assert 0 < total <= naive, (name, total, naive)
        if name != "mlp_layers_float32":      # (every intermediate of that plan is an output)
            assert total < naive, (name, total, naive)


def test_replayed_negative_stride_output_is_refreshed():
    """An output that is a reversed view is materialised INSIDE the recorded launches: replays
    with new input values return the new reversed data (not the first call's copy)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan
    p = Plan("rev", {}, [], [], [])
    x = p.new_var("float64", [None], "x")
    y = p.new_var("float64", [None])
    o = p.new_var("float64", [None])
    p.inputs, p.outputs = [x], [o]
    sc = {"n_in": 1, "nodes": [{"op": "exp", "in": [["i", 0]], "dtype": "float64"}], "out": [["t", 0]]}
    p.nodes = [Node("Elemwise", [x], [y], {"scalar": sc}),
               Node("Subtensor", [y], [o], {"idx_list": [{"slice": [None, None, -1]}]})]
    ex = PlanExecutor(p, use_graph=True)
    xd = torch.zeros(100, dtype=torch.float64, device="cuda")
    for k in range(4):
        xd.copy_(torch.arange(100, dtype=torch.float64, device="cuda") * 0.01 * (k + 1))
        (got,) = ex(xd)
        np.testing.assert_allclose(got.cpu().numpy(), np.exp(xd.cpu().numpy())[::-1], rtol=1e-14)


def test_replay_rebinds_to_fresh_device_tensors_without_staging():
    """New device tensors of a known layout (a training loop's next batch) replay the recorded
    launches with rebound addresses (ahip_list_run_rebased): no staging copies, same results as
    the eager path; overlapping / differently aligned buffers fall back safely."""
    import torch
    from golden_util import CASES, case_plan
    from aesara_amd.executor import PlanExecutor
    c = next(c for c in CASES if c["name"] == "nll_classifier_float32")
    eager = PlanExecutor(case_plan(c))
    replay = PlanExecutor(case_plan(c), use_graph=True)
    rng = np.random.default_rng(0)
    keep = []

    def batch():
        x = rng.standard_normal((48, 20)).astype("float32")
        W = (rng.standard_normal((20, 10)) * 0.5).astype("float32")
        b = (rng.standard_normal(10) * 0.1).astype("float32")
        y = rng.integers(0, 10, 48)
        ts = [torch.from_numpy(v).cuda() for v in (x, W, b, y)]
        keep.append(ts)                      # keep them alive: every batch has new addresses
        return ts

    for k in range(8):
        ins = batch()
        want = [o.cpu().numpy() for o in eager(*ins)]
        got = [o.cpu().numpy() for o in replay(*ins)]
        for g, w in zip(got, want):
            np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-7)
    assert len(replay._stage) == 0, "device batches must be rebound, not staged"
    assert len(replay._reloc) == 1 and len(replay._list_refs) == 1
    # a misaligned view of the same layout is a different signature (kernels were chosen for the
    # alignment class): it gets its own recording, results stay right
    big = torch.zeros(48 * 20 + 1, dtype=torch.float32, device="cuda")
    ins = batch()
    big[1:].copy_(ins[0].reshape(-1))
    xs = big[1:].view(48, 20)
    want = [o.cpu().numpy() for o in eager(xs, *ins[1:])]
    got = [o.cpu().numpy() for o in replay(xs, *ins[1:])]
    for g, w in zip(got, want):
        np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-7)
    # the same tensor passed for two inputs that were distinct when recorded is not rebound
    c2 = next(c for c in CASES if c["name"] == "cfg1b_matrix_add")
    r2 = PlanExecutor(case_plan(c2), use_graph=True)
    a = torch.ones(32, 32, dtype=torch.float64, device="cuda")
    b2 = torch.full((32, 32), 2.0, dtype=torch.float64, device="cuda")
    assert float(r2(a, b2)[0][0, 0]) == 3.0
    c3 = torch.full((32, 32), 5.0, dtype=torch.float64, device="cuda")
    assert float(r2(c3, b2)[0][0, 0]) == 7.0          # rebound
    assert float(r2(c3, c3)[0][0, 0]) == 10.0         # aliased inputs: safe path
    assert float(r2(a, c3)[0][0, 0]) == 6.0


@pytest.mark.parametrize("borrow", [False, True])
def test_replay_output_fed_back_as_input(borrow):
    """An iteration h = f(h, W) hands each result straight back as the next call's input.  With
    borrowed outputs that input IS the function-owned output buffer of the recorded launches (and
    with fresh outputs it is a buffer the list once wrote): the replay must notice the overlap and
    still compute tanh(h @ W) of the OLD h — checked against the eager executor, 12 rounds."""
    import torch
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan, Var
    tanh = {"n_in": 1, "nodes": [{"op": "tanh", "in": [["i", 0]], "dtype": "float32"}], "out": [["t", 0]]}
    vs = {0: Var(0, "float32", [None, None]), 1: Var(1, "float32", [None, None]),
          2: Var(2, "float32", [None, None]), 3: Var(3, "float32", [None, None])}
    plan = Plan("feedback", vs, [0, 1], [3], [Node("Dot22", [0, 1], [2], {}),
                                              Node("Elemwise", [2], [3], {"scalar": tanh})])
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    W = torch.randn(96, 96, device="cuda", generator=g) * 0.3
    h0 = torch.randn(64, 96, device="cuda", generator=g)
    eager = PlanExecutor(plan)
    replay = PlanExecutor(plan, use_graph=True, borrow=borrow)
    he, hr = h0.clone(), h0.clone()
    for k in range(12):
        (he,) = eager(he, W)
        (hr,) = replay(hr, W)
        assert torch.allclose(hr, he, rtol=1e-5, atol=1e-6), k
    # the transposed result fed back (a view of the previous output with other strides)
    Wt = W.t().contiguous()
    (a,) = eager(he.t().contiguous().t(), Wt)
    (b,) = replay(hr.t().contiguous().t(), Wt)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
