"""A Scan whose step is purely element-wise as ONE kernel launch (aesara_amd/scan_persist_ew.py:
thread e runs the whole recurrence of element e; taps in registers, sequences read ahead, a
do-while's stop decided on the device): parity with the reference's outputs (goldens), with the
launch-list path, and with NumPy restatements at sizes far beyond the goldens'."""
import numpy as np
import pytest

from golden_util import CASES, assert_matches, case_expected, case_inputs, case_plan

pytestmark = pytest.mark.gpu

ELEMENTWISE = ["scan_cumsum", "scan_taps", "scan_two_outputs", "scan_while_cumsum", "scan_while_never_stops"]
NO_RECURRENCE = ["scan_nitsot_map", "scan_map_jacobian_rows", "scan_map_hessian_unit_vectors",
                 "scan_map_rows_reduce_broadcast"]


def _case(name):
    return next(c for c in CASES if c["name"] == name)


def _np(outs):
    return [o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o) for o in outs]


@pytest.mark.parametrize("name", ELEMENTWISE)
@pytest.mark.parametrize("use_graph", [False, True])
def test_elementwise_scans_run_as_one_launch_and_match(name, use_graph):
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(3):
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"call {it}")
    assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
    assert "element-wise" in list(ex.scan_notes.values())[0]
    E.TUNE["scan_persist"] = 0
    try:
        ex2 = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        ref = _np(ex2(*ins))
        assert all(v.startswith("launch-list") for v in ex2.scan_modes.values())
    finally:
        E.TUNE["scan_persist"] = 1
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r)          # the same arithmetic, element by element
    ex.check()


@pytest.mark.parametrize("name", NO_RECURRENCE)
@pytest.mark.parametrize("use_graph", [False, True])
def test_scans_without_recurrence_are_one_evaluation_over_whole_sequences(name, use_graph):
    """aesara.map / the row loops of gradient.jacobian and gradient.hessian (scan/basic.py:71 without
    outputs_info; gradient.py:1930, :2027): no step reads what another wrote, so the step plan is
    restated over whole sequences (fusion.batch_map_step) and evaluated once — results = the
    reference's outputs = the launch-list path (to the goldens' tolerance: a Gemv per step there,
    one GEMM here)."""
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(3):
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"call {it}")
    assert set(ex.scan_modes.values()) == {"all-rows"}, ex.scan_modes
    E.TUNE["scan_persist"] = 0
    try:
        ex2 = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        ref = _np(ex2(*ins))
        assert all(v.startswith("launch-list") for v in ex2.scan_modes.values())
    finally:
        E.TUNE["scan_persist"] = 1
    assert_matches(c, ref, case_expected(c), "launch list")
    ex.check()


def test_jacobian_rows_at_size_and_index_errors():
    """The golden Jacobian plan at n = 700 (700 steps -> one evaluation) against NumPy, outputs that
    own their buffers across calls, and a step index out of range raising the reference's
    IndexError (subtensor.py:756) from the whole-sequence gather."""
    import torch
    from aesara_amd import executor as E
    c = _case("scan_map_jacobian_rows")
    ex = E.PlanExecutor(case_plan(c))
    n = 700
    rng = np.random.default_rng(3)
    x, W = rng.standard_normal(n), rng.standard_normal((n, n)) / np.sqrt(n)
    (J,) = _np(ex(torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda()))
    t = np.tanh(W @ x)
    want = ((1 - t * t) * x.sum())[:, None] * W + t[:, None] * np.ones(n)[None, :]
    np.testing.assert_allclose(J, want, rtol=1e-10, atol=1e-12)
    assert set(ex.scan_modes.values()) == {"all-rows"}
    assert "once" in list(ex.scan_notes.values())[0]
    # the same evaluation with a byte budget of 1 MiB for its intermediates (7 x [700, 700] float64 = 27 MB
    # all at once): the rows go in blocks, the result is the same
    import os
    os.environ["AESARA_HIP_ROWS_BYTES_CAP_GB"] = str(1.0 / 1024)
    try:
        for use_graph in (False, True):
            exb = E.PlanExecutor(case_plan(c), use_graph=use_graph)
            for _ in range(2):
                (Jb,) = _np(exb(torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda()))
                np.testing.assert_array_equal(Jb, J)
            assert "blocks of" in list(exb.scan_notes.values())[0], exb.scan_notes
    finally:
        del os.environ["AESARA_HIP_ROWS_BYTES_CAP_GB"]
    c2 = _case("scan_map_rows_reduce_broadcast")
    ex2 = E.PlanExecutor(case_plan(c2))
    M, b, idx = case_inputs(c2)
    bad = np.array(idx, copy=True)
    bad[4] = M.shape[0]
    with pytest.raises(IndexError):
        ex2(M, b, bad)
        ex2.check()
    got = _np(ex2(M, b, idx))
    assert_matches(c2, got, case_expected(c2), "after the failed call")


@pytest.mark.parametrize("name", ["scan_grad_taps", "scan_grad_taps_wide", "scan_grad_taps13"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_gradient_scans_with_mit_mot_windows_run_as_one_launch(name, use_graph):
    """Scan.L_op's gradient Scans of element-wise recurrences with several taps: mit-mot outputs
    with taps [0, 2, 1] -> [2, 1] / [0, 3, 1] -> [3, 1] (scan/op.py:2379; scan_perform.pyx:343-352
    reads rows t + tap, :437-452 writes rows t + out-tap).  Forward and gradient Scan both run as one
    launch; results = the reference's outputs = the launch-list path bit for bit."""
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(3):
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"call {it}")
    assert list(ex.scan_modes.values()) == ["persistent", "persistent"], ex.scan_modes
    E.TUNE["scan_persist"] = 0
    try:
        ex2 = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        ref = _np(ex2(*ins))
        assert all(v.startswith("launch-list") for v in ex2.scan_modes.values())
    finally:
        E.TUNE["scan_persist"] = 1
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r)
    ex.check()


@pytest.mark.parametrize("T,n", [(700, 3000), (1, 5), (9, 1), (257, 70001)])
def test_cumsum_recurrence_large(T, n):
    """s_t = s_{t-1} + x_t over [T, n]: every row against np.cumsum (fp64, same order: exact)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(T + n)
    x, s0 = rng.standard_normal((T, n)), rng.standard_normal(n)
    ex = PlanExecutor(case_plan(_case("scan_cumsum")))
    outs = _np(ex(torch.from_numpy(x).cuda(), torch.from_numpy(s0).cuda()))
    want = np.cumsum(np.vstack([s0[None], x]), axis=0)[1:]        # ((s0 + x_0) + x_1) + ...: the same order
    got = outs[0]
    assert got.shape[0] >= T
    np.testing.assert_array_equal(got[-T:], want)
    assert list(ex.scan_modes.values()) == ["persistent"]


def test_two_tap_recurrence_long():
    """y_t = w * y_{t-2} + 0.5 * y_{t-1} + x_t (golden scan_taps' step) for 5000 steps."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(7)
    T = 5000
    x, init, w = rng.standard_normal(T) * 0.1, rng.standard_normal(2), np.float64(0.3)
    ex = PlanExecutor(case_plan(_case("scan_taps")))
    (res,) = _np(ex(torch.from_numpy(x).cuda(), torch.from_numpy(init).cuda(), w))
    a, b = init
    want = []
    for t in range(T):
        y = w * a + 0.5 * b + x[t]
        want.append(y)
        a, b = b, y
    np.testing.assert_allclose(res[-T:], np.array(want), rtol=1e-13, atol=1e-13)
    assert list(ex.scan_modes.values()) == ["persistent"]


@pytest.mark.parametrize("stop_at", [1, 37, 999, None])
def test_do_while_stops_on_the_device(stop_at):
    """s_t = s_{t-1} + x_t until s_t > 3 (golden scan_while_cumsum's step) over 1000 steps: the
    trip count, the truncated output and the last value against a NumPy loop; one launch, one
    host read (eager: a do-while plan is never replayed)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    T = 1000
    x = np.full(T, 1e-3)
    if stop_at is not None:
        x[stop_at - 1] = 5.0                 # the step whose sum first exceeds 3
    ex = PlanExecutor(case_plan(_case("scan_while_cumsum")), use_graph=True)
    for call in range(2):
        res, last, count = _np(ex(torch.from_numpy(x).cuda()))
        k = T if stop_at is None else stop_at
        assert int(count) == k and res.shape == (k,)
        np.testing.assert_allclose(res, np.cumsum(x)[:k], rtol=1e-13)
        np.testing.assert_allclose(last, np.cumsum(x)[k - 1], rtol=1e-13)
    assert list(ex.scan_modes.values()) == ["persistent"]


# ---- full reductions inside the step: ONE workgroup, thread e = element e (round 5) ------------
REDUCTIONS = ["scan_red_normalise", "scan_red_normalise_wide_f32", "scan_red_until_all",
              "scan_red_newton_until", "scan_red_energy", "scan_red_running_total", "scan_red_int_minmax"]


@pytest.mark.parametrize("name", REDUCTIONS)
@pytest.mark.parametrize("use_graph", [False, True])
def test_steps_with_full_reductions_run_as_one_launch(name, use_graph):
    """Steps holding full reductions (scan_perform.pyx:309-541 runs them in the same loop; a
    do-while over a VECTOR state is one, :424-426): one launch, results = the reference's outputs
    (integer case: exact) and = the launch-list path (reductions fold in another order there:
    compared at the case's tolerance, integers bit for bit)."""
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(3):
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"call {it}")
    assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
    assert "element-wise" in list(ex.scan_notes.values())[0]
    E.TUNE["scan_persist"] = 0
    try:
        ex2 = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        ref = _np(ex2(*ins))
        assert all(v.startswith("launch-list") for v in ex2.scan_modes.values())
    finally:
        E.TUNE["scan_persist"] = 1
    assert_matches(c, got, ref, "against the launch-list path")
    ex.check()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 700, 1024, 1025, 5000])
def test_normalising_recurrence_over_sizes(n):
    """x <- x * a / sum|x * a| for 12 steps at sizes around the wavefront / workgroup limits (one
    partial wavefront, exactly one, several, the full 1024-thread workgroup) against NumPy; beyond
    1024 elements the loop runs on the launch list (same results)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(n)
    x0, a = rng.uniform(0.5, 1.5, n), rng.uniform(0.5, 1.5, n)
    ex = PlanExecutor(case_plan(_case("scan_red_normalise")), use_graph=True)
    for call in range(2):
        res, last = _np(ex(torch.from_numpy(x0).cuda(), torch.from_numpy(a).cuda()))
        x, want = x0, []
        for _ in range(7):
            x = x * a / np.abs(x * a).sum()
            want.append(x)
        np.testing.assert_allclose(res[-7:], np.array(want), rtol=1e-12)
        np.testing.assert_allclose(last, want[-1], rtol=1e-12)
    mode = list(ex.scan_modes.values())[0]
    assert (mode == "persistent") == (n <= 1024), mode
    ex.check()


@pytest.mark.parametrize("rows,cols,thr", [(15, 5, 5.0), (40, 257, 11.0), (9, 1000, 100.0), (30, 64, -1.0)])
def test_vector_do_while_stops_on_the_device(rows, cols, thr):
    """``until(at_all(x > u))`` over matrix rows (tests/scan/test_basic.py:2391
    test_grad_until_ndim_greater_one): the condition is a reduction of a vector, every thread of
    the workgroup reaches the same decision; trip count and truncated output against NumPy
    (a threshold no row passes: the loop runs to the end; one the first row passes: one step)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    X = np.tile(np.arange(rows, dtype="float64").reshape(-1, 1), (1, cols))
    X[:, -1] -= 0.5                                   # the last column passes the threshold one row later
    ex = PlanExecutor(case_plan(_case("scan_red_until_all")), use_graph=True)
    stop = next((i for i in range(rows) if np.all(X[i] > thr)), rows - 1)
    for call in range(2):
        res, count = _np(ex(torch.from_numpy(X).cuda(), np.float64(thr)))
        assert int(count) == stop + 1 and res.shape == (stop + 1, cols)
        np.testing.assert_array_equal(res, (X * X)[:stop + 1])
    assert list(ex.scan_modes.values()) == ["persistent"]
    ex.check()
