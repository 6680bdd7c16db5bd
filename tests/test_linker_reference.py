"""The drop-in boundary against the REAL reference front end (authoring container only; skipped
on the GPU box where /root/reference is absent): HipLinker plugs into aesara.function through
the JITLinker API, and the plans committed under tests/golden are exactly what it lowers."""
import json

import numpy as np
import pytest

import ref_overlay

pytestmark = pytest.mark.skipif(not ref_overlay.available(),
                                reason="reference Aesara not present (GPU box)")


@pytest.fixture(scope="module")
def ae():
    return ref_overlay.import_reference()


def _oracle_linker():
    import interp
    from aesara_amd.linker import HipLinker
    return HipLinker(executor_factory=lambda plan: (lambda *a: interp.run_plan(plan, a)))


def test_mode_registration(ae):
    from aesara.compile.mode import get_mode, predefined_linkers
    import aesara_amd
    mode = aesara_amd.get_mode()
    assert "hip" in predefined_linkers
    assert get_mode("HIP") is mode
    from aesara.link.basic import JITLinker
    assert isinstance(mode.linker, JITLinker)


def test_function_through_jitlinker_matches_c_linker(ae):
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    y = at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum()
    f_hip = ae.function([x, mu, sg], y, mode=Mode(_oracle_linker(), HIP_QUERY))
    f_ref = ae.function([x, mu, sg], y)  # reference default: cvm + C thunks
    xv = np.random.default_rng(1).standard_normal((64, 48))
    np.testing.assert_allclose(f_hip(xv, 0.1, 1.3), f_ref(xv, 0.1, 1.3), rtol=1e-12)
    # README example (README.md:49-60): a + b on dscalars
    a, b = at.dscalar("a"), at.dscalar("b")
    f = ae.function([a, b], a + b, mode=Mode(_oracle_linker(), HIP_QUERY))
    assert f(1.5, 2.5) == 4.0


def test_shared_variable_updates(ae):
    """check_blas.py:54-57 pattern: updates=[(C, 0.4*C + 0.8*dot(A,B))] — the linker returns the
    update expression as an extra output and Function stores it back."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    rng = np.random.default_rng(0)
    A, B = (ae.shared(rng.standard_normal((8, 8)).astype("float32")) for _ in range(2))
    Cs = ae.shared(np.zeros((8, 8), "float32"))
    f = ae.function([], [], updates=[(Cs, np.float32(0.4) * Cs + np.float32(0.8) * at.dot(A, B))],
                    mode=Mode(_oracle_linker(), HIP_QUERY))
    f(); f()
    ab = A.get_value() @ B.get_value()
    np.testing.assert_allclose(Cs.get_value(), 0.4 * 0.8 * ab + 0.8 * ab, rtol=1e-5)
    ops = [n.op for n in f.maker.linker.plan.nodes]
    assert ops == ["Gemm"]


def test_device_shared_variable_state_stays_on_device(ae):
    """with ``hip_shared`` the check_blas update pattern never converts the
    state to a host array: the storage cell holds a torch tensor before and after each call,
    the thunk receives it as is, and the update output is stored back as is."""
    import torch
    import aesara.tensor as at
    import interp
    from aesara.compile.mode import Mode
    from aesara.tensor.type import TensorType
    from aesara_amd.linker import HIP_QUERY, HipLinker
    from aesara_amd.sharedvar import hip_shared
    seen = []

    def factory(plan):
        def run(*a):
            seen.append([type(x) for x in a])
            outs = interp.run_plan(plan, [x.numpy() if isinstance(x, torch.Tensor) else x
                                          for x in a])
            return [torch.from_numpy(np.ascontiguousarray(o)) for o in outs]
        return run

    rng = np.random.default_rng(0)
    Av, Bv = (rng.standard_normal((8, 8)).astype("float32") for _ in range(2))
    A, B = hip_shared(Av, device="cpu"), hip_shared(Bv, device="cpu")
    C = hip_shared(np.zeros((8, 8), "float32"), name="C", device="cpu")
    assert type(C.type) is TensorType and isinstance(C.container.value, torch.Tensor)
    f = ae.function([], [], updates=[(C, np.float32(0.4) * C + np.float32(0.8) * at.dot(A, B))],
                    mode=Mode(HipLinker(executor_factory=factory), HIP_QUERY))
    want = np.zeros((8, 8), "float32")
    for _ in range(3):
        f()
        want = np.float32(0.4) * want + np.float32(0.8) * (Av @ Bv)
        assert isinstance(C.container.value, torch.Tensor)       # never became an ndarray
    assert len(seen) == 3 and all(t is torch.Tensor for call in seen for t in call)
    got = C.get_value()
    assert isinstance(got, np.ndarray)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    # host values are filtered by the reference's TensorType.filter, then uploaded
    C.set_value(np.ones((8, 8), "float32"))
    assert isinstance(C.get_value(borrow=True, return_internal_type=True), torch.Tensor)
    with pytest.raises(TypeError):
        C.set_value(torch.zeros(8, dtype=torch.float32))            # rank
    with pytest.raises(TypeError):
        C.set_value(torch.zeros((8, 8), dtype=torch.float64))       # dtype
    # explicit inputs: device tensors pass with trust_input (reference types.py:851-856)
    x = at.dvector("x")
    g = ae.function([x], (x * 2.0).sum(),
                    mode=Mode(HipLinker(executor_factory=factory), HIP_QUERY))
    g.trust_input = True
    assert float(g(torch.arange(5, dtype=torch.float64))) == 20.0


def test_mlp_training_step_through_the_linker_matches_c_linker(ae):
    """A whole SGD step of a 2-layer softmax classifier (Gemm, row-chain log-softmax, N-d
    integer indexing for the NLL, its gradient scatter, Argmax, four shared-variable updates)
    compiled with mode=HIP and run for several steps: losses, predictions and parameters follow
    the reference's C linker; the parameters live in device-resident shared variables."""
    import torch
    import aesara.tensor as at
    import interp
    from aesara.compile.mode import Mode
    from aesara.tensor.special import log_softmax
    from aesara_amd.linker import HIP_QUERY, HipLinker
    from aesara_amd.sharedvar import hip_shared

    def factory(plan):
        def run(*a):
            outs = interp.run_plan(plan, [x.numpy() if isinstance(x, torch.Tensor) else x
                                          for x in a])
            return [torch.from_numpy(np.ascontiguousarray(o)) for o in outs]
        return run

    rng = np.random.default_rng(0)
    init = [rng.standard_normal((20, 32)) * 0.3, np.zeros(32), rng.standard_normal((32, 10)) * 0.3,
            np.zeros(10)]

    def build(make_shared, mode):
        x, y = at.dmatrix("x"), at.lvector("y")
        W1, b1, W2, b2 = params = [make_shared(v.copy()) for v in init]
        logits = at.dot(at.tanh(at.dot(x, W1) + b1), W2) + b2
        loss = -log_softmax(logits, axis=-1)[at.arange(y.shape[0]), y].mean()
        upd = [(p, p - 0.1 * g) for p, g in zip(params, ae.grad(loss, params))]
        return ae.function([x, y], [loss, at.argmax(logits, axis=1)], updates=upd, mode=mode), params

    f_hip, p_hip = build(lambda v: hip_shared(v, device="cpu"),
                         Mode(HipLinker(executor_factory=factory, return_numpy=True), HIP_QUERY))
    f_ref, p_ref = build(ae.shared, None)
    for step in range(4):
        xv = rng.standard_normal((64, 20))
        yv = rng.integers(0, 10, 64)
        l1, a1 = f_hip(xv, yv)
        l2, a2 = f_ref(xv, yv)
        np.testing.assert_allclose(l1, l2, rtol=1e-10)
        np.testing.assert_array_equal(a1, a2)
    for ph, pr in zip(p_hip, p_ref):
        assert isinstance(ph.container.value, torch.Tensor)
        np.testing.assert_allclose(ph.get_value(), pr.get_value(), rtol=1e-9, atol=1e-12)


def _dry_linker(**kw):
    """HipLinker over the REAL PlanExecutor in dry-run mode (host logic, no device): the error
    and profile plumbing between executor steps and Apply nodes runs exactly as on the GPU."""
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.linker import HipLinker

    class DryHipLinker(HipLinker):
        def jit_compile(self, plan):
            ex = PlanExecutor(plan, dry_run=True)
            if self.profile:
                ex.enable_profile()
            self.executor = ex
            return ex
    return DryHipLinker(**kw)


def test_errors_are_annotated_with_the_apply_node(ae):
    """``Function.__call__`` re-raises through ``raise_with_op`` (link/utils.py:270) with the
    Apply node named by ``vm.position_of_error`` (types.py:974-991); the exception keeps its
    type (tests/tensor/test_elemwise.py:769-792 match on ValueError)."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    a, b, v = at.dmatrix("a"), at.dmatrix("b"), at.dvector("v")
    f = ae.function([a, b, v], [at.dot(a, v).sum(), at.exp(a) + b * 2.0],
                    mode=Mode(_dry_linker(), HIP_QUERY))
    with pytest.raises(ValueError) as ei:
        f(np.zeros((3, 4)), np.zeros((5, 4)), np.zeros(4))
    msg = str(ei.value)
    assert "Shapes on dimension 0 do not match" in msg
    assert "Apply node that caused the error: Elemwise{Composite" in msg
    assert "Inputs shapes: [(3, 4), (1, 1), (5, 4)]" in msg and "[HIP step" in msg
    with pytest.raises(ValueError) as ei:          # the OTHER node of the same function
        f(np.zeros((3, 4)), np.zeros((3, 4)), np.zeros(7))
    line = next(ln for ln in str(ei.value).splitlines() if ln.startswith("Apply node that caused"))
    assert "Gemv" in line or "dot" in line.lower(), line
    # a later valid call is unaffected and the cells of non-borrowed outputs were dropped
    assert len(f(np.zeros((3, 4)), np.zeros((3, 4)), np.zeros(4))) == 2


def test_profile_stats_are_filled_per_apply_node(ae):
    """``aesara.function(profile=True)``: the linker keeps the ProfileStats it ``accept``s
    (link/vm.py:868) and ``vm.update_profile`` (link/vm.py:251) books step times on Apply nodes."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    x, v = at.dmatrix("x"), at.dvector("v")
    f = ae.function([x, v], [at.dot(x, v).sum(), at.exp(x).sum(axis=0)],
                    mode=Mode(_dry_linker(), HIP_QUERY), profile=True)
    assert f.maker.linker.profile is f.profile and f.maker.linker.executor.profiling
    for _ in range(4):
        f(np.ones((40, 30)), np.arange(30.0))
    prof = f.profile
    assert prof.fct_callcount >= 3 and prof.vm_call_time > 0
    assert prof.apply_time and sum(prof.apply_time.values()) > 0
    assert all(c >= 3 for c in prof.apply_callcount.values())
    # fused steps are booked on the node that produces their result: here the two Sums
    topo = f.maker.fgraph.toposort()
    assert len(prof.apply_time) >= 2 and all(n in topo for (_fg, n) in prof.apply_time)
    assert all(fg is f.maker.fgraph for (fg, _n) in prof.apply_time)


def test_accept_keeps_linker_options(ae):
    """A bound linker asked to accept another graph (PerformLinker.accept re-creates itself from
    allow_gc alone, link/basic.py:319-322) keeps return_numpy / use_graph / fast_call."""
    import aesara.tensor as at
    from aesara.graph.fg import FunctionGraph
    from aesara_amd.linker import HipLinker
    x = at.dvector("x")
    l1 = HipLinker(return_numpy=True, use_graph=False).accept(FunctionGraph([x], [x + 1.0]))
    y = at.dvector("y")
    l2 = l1.accept(FunctionGraph([y], [y * 2.0]))
    assert l2 is not l1 and l2.return_numpy and not l2.use_graph and l2.fgraph is not l1.fgraph


def test_mixed_dtype_and_integer_dot_are_lowered_float16_is_refused_at_compile_time(ae):
    """No dtype surprise at run time: the operands of a ``Dot`` are cast to the node's output
    dtype when it is lowered (``Dot.make_node`` upcasts, ``np.dot`` converts to the common type:
    tensor/math.py:1903 / :1929) — float32 x float64 -> float64, int32 x float32 -> float64,
    int8 x int64 -> int64 — and an integer product keeps NumPy's wrap-around arithmetic; a
    float16 ``Dot`` (no kernel) raises ``UnsupportedOp`` when the function is compiled."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    from aesara_amd.lower import UnsupportedOp
    x, y = at.fmatrix("x"), at.dmatrix("y")
    f = ae.function([x, y], [at.dot(x, y), at.dot(y.T, x.T)], mode=Mode(_oracle_linker(), HIP_QUERY))
    xv = np.random.default_rng(0).standard_normal((4, 5)).astype("float32")
    yv = np.random.default_rng(1).standard_normal((5, 3))
    r1, r2 = f(xv, yv)
    assert r1.dtype == np.float64
    np.testing.assert_allclose(r1, xv @ yv, rtol=1e-12)
    np.testing.assert_allclose(r2, yv.T @ xv.T, rtol=1e-12)
    i, j, b = at.lmatrix("i"), at.lmatrix("j"), at.bmatrix("b")
    g = ae.function([i, j, b, x], [at.dot(i, j), at.dot(b, b.T), at.dot(i, x.T)],
                    mode=Mode(_oracle_linker(), HIP_QUERY))
    rng = np.random.default_rng(2)
    iv, jv = rng.integers(-2 ** 40, 2 ** 40, (4, 5)), rng.integers(-2 ** 40, 2 ** 40, (5, 4))
    bv = rng.integers(-128, 127, (4, 5)).astype("int8")
    o1, o2, o3 = g(iv, jv, bv, xv)
    assert o1.dtype == np.int64 and o2.dtype == np.int8 and o3.dtype == np.float64
    np.testing.assert_array_equal(o1, iv @ jv)                  # wraps like NumPy
    np.testing.assert_array_equal(o2, bv @ bv.T)
    np.testing.assert_allclose(o3, iv @ xv.T.astype("float64"), rtol=1e-12)
    h1, h2 = at.matrix("h1", dtype="float16"), at.matrix("h2", dtype="float16")
    with pytest.raises(UnsupportedOp, match="float16"):
        ae.function([h1, h2], at.dot(h1, h2), mode=Mode(_oracle_linker(), HIP_QUERY))


def test_linker_clone_and_scan_inner_mode(ae):
    """Linker.clone(allow_gc=…) is used by Scan/Mode.clone (link/basic.py:190)."""
    from aesara_amd.linker import HipLinker
    l2 = HipLinker().clone(allow_gc=False)
    assert isinstance(l2, HipLinker) and l2.allow_gc is False


def test_unsupported_op_fails_loudly(ae):
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    from aesara_amd.lower import UnsupportedOp
    x = at.dmatrix("x")
    with pytest.raises(UnsupportedOp):
        ae.function([x], at.slinalg.cholesky(at.dot(x, x.T)), mode=Mode(_oracle_linker(), HIP_QUERY))


def test_committed_plans_are_what_the_linker_lowers(ae):
    """Re-lower EVERY golden graph with the live reference and compare with the committed plan
    JSON (the GPU box executes the committed plans: they must be what the linker produces)."""
    import gen_golden
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    from golden_util import CASES
    committed = {c["name"]: c["plan"] for c in CASES}
    assert set(committed) == {n for n, *_ in gen_golden.CASES}
    bad = []
    for name, fn, *_ in gen_golden.CASES:
        ins, outs, _specs = fn()
        f = ae.function(ins, outs, mode=Mode(_oracle_linker(), HIP_QUERY), on_unused_input="ignore",
                        accept_inplace=True)
        plan = f.maker.linker.plan
        plan.name = name
        if json.dumps(plan.to_json(), sort_keys=True) != json.dumps(committed[name], sort_keys=True):
            bad.append(name)
    assert not bad, bad


def _picklable_oracle_factory(plan):
    import interp
    return lambda *a: interp.run_plan(plan, a)


def test_function_api_surface_clone_copy_swap_pickle_profile(ae):
    """What `Function` / `Mode` do to a linker besides calling it: `Mode.clone` with
    link kwargs (`Linker.clone`, link/basic.py:190), `Function.copy(swap=...)` (types.py:558),
    pickling (re-links on load, types.py:1125), `profile=True`."""
    import pickle
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY, HipLinker
    m = Mode(HipLinker(executor_factory=_picklable_oracle_factory, return_numpy=True), HIP_QUERY)
    m2 = m.clone(link_kwargs=dict(allow_gc=False))
    assert isinstance(m2.linker, HipLinker) and m2.linker.allow_gc is False and m2.linker.return_numpy
    x = at.dvector("x")
    f = ae.function([x], (x * 2).sum(), mode=m2)
    assert f(np.arange(4.0)) == 12.0 and f.copy()(np.arange(5.0)) == 20.0
    s = ae.shared(np.ones(3))
    g = ae.function([], s * 2, mode=m)
    np.testing.assert_array_equal(g.copy(swap={s: ae.shared(np.full(3, 5.0))})(), [10.0] * 3)
    f2 = pickle.loads(pickle.dumps(f))
    assert f2(np.arange(3.0)) == 6.0
    fp = ae.function([x], (x ** 2).sum(), mode=m, profile=True)
    assert fp(np.arange(4.0)) == 14.0 and fp.profile.fct_call_time > 0


def test_large_graph_constants_are_kept_by_reference(ae):
    """Data embedded in the graph (`at.as_tensor(X)` of a design matrix): the plan holds the array
    itself, serialises it only on demand, and the executor uploads it once."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.linker import HIP_QUERY
    from aesara_amd.plan import Plan
    import interp
    X = np.random.default_rng(0).standard_normal((200, 50))
    w = at.dvector("w")
    out = at.dot(at.as_tensor(X), w).sum() + (at.as_tensor(X) ** 2).sum(axis=0)[:3].sum()
    f = ae.function([w], out, mode=Mode(_oracle_linker(), HIP_QUERY))
    wv = np.arange(50.0)
    np.testing.assert_allclose(f(wv), (X @ wv).sum() + (X ** 2).sum(axis=0)[:3].sum(), rtol=1e-12)
    plan = f.maker.linker.plan
    big = [v for v in plan.vars.values() if v.const is not None and "array" in v.const]
    assert len(big) == 1 and big[0].const["array"].shape == (200, 50)
    np.testing.assert_allclose(interp.run_plan(Plan.loads(plan.dumps()), [wv])[0], f(wv), rtol=1e-12)
    PlanExecutor(plan, dry_run=True)(wv)


def test_rnn_language_model_adam_step_matches_reference(ae):
    """Embedding gather -> Scan (tanh RNN over a batch) -> softmax cross-entropy (tensor.nnet) ->
    `aesara.grad` -> Adam updates of six parameters with their moment buffers (19 shared-variable
    updates), three consecutive steps: losses and parameters follow the reference's linker."""
    import aesara.tensor as at
    import aesara.tensor.nnet as nn
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY, HipLinker
    V, E, H, T, B = 30, 8, 12, 7, 5

    def build(mode):
        r0 = np.random.default_rng(1)
        params = [ae.shared(r0.standard_normal(s) * 0.2)
                  for s in ((V, E), (E, H), (H, H), (H,), (H, V), (V,))]
        Wemb, Wx, Wh, bh, Wo, bo = params
        tok, tgt = at.lmatrix("tok"), at.lmatrix("tgt")
        emb = Wemb[tok.flatten()].reshape((T, B, E))
        hs, _ = ae.scan(lambda x_t, h: at.tanh(at.dot(x_t, Wx) + at.dot(h, Wh) + bh),
                        sequences=[emb], outputs_info=[at.zeros((B, H))])
        logits = at.dot(hs.reshape((T * B, H)), Wo) + bo
        loss = nn.categorical_crossentropy(nn.softmax(logits), tgt.flatten()).mean()
        t = ae.shared(np.float64(0.0))
        ups = [(t, t + 1)]
        for p, g in zip(params, ae.grad(loss, params)):
            m, v = (ae.shared(np.zeros(p.get_value().shape)) for _ in range(2))
            m2, v2 = 0.9 * m + 0.1 * g, 0.999 * v + 0.001 * g * g
            mh, vh = m2 / (1 - 0.9 ** (t + 1)), v2 / (1 - 0.999 ** (t + 1))
            ups += [(m, m2), (v, v2), (p, p - 0.01 * mh / (at.sqrt(vh) + 1e-8))]
        return ae.function([tok, tgt], loss, updates=ups, mode=mode), params

    f_hip, p_hip = build(Mode(HipLinker(executor_factory=_picklable_oracle_factory, return_numpy=True),
                              HIP_QUERY))
    f_ref, p_ref = build(Mode("py", "fast_run"))
    rng = np.random.default_rng(0)
    for _ in range(3):
        tok, tgt = rng.integers(0, V, (T, B)), rng.integers(0, V, (T, B))
        np.testing.assert_allclose(f_hip(tok, tgt), f_ref(tok, tgt), rtol=1e-10)
    for a, b in zip(p_hip, p_ref):
        np.testing.assert_allclose(a.get_value(), b.get_value(), rtol=1e-8, atol=1e-10)
    from aesara_amd.executor import PlanExecutor
    PlanExecutor(f_hip.maker.linker.plan, dry_run=True)     # every step has a kernel / launch plan


def test_function_drives_the_real_executor_host_path(ae):
    """`aesara.function(..., mode=HIP)` over the REAL `PlanExecutor` in dry-run mode (no device:
    every C-ABI call is recorded instead of issued): `Function.__call__` -> fast VM -> executor
    host logic (binding, shapes, fusion, kernel generation + hiprtc compilation, launch
    recording), for BASELINE configs 2, 3b, 4 and 5; slow (`streamline`) and fast VMs agree."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.device import DevArray
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.linker import HIP_QUERY, HipLinker

    execs = []

    def factory(plan):
        ex = PlanExecutor(plan, dry_run=True)
        execs.append(ex)
        return ex

    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    A, B = at.fmatrix("A"), at.fmatrix("B")
    X, w, b, yv = at.fmatrix("X"), at.fvector("w"), at.fscalar("b"), at.fvector("y")
    z = at.dot(X, w) + b
    logp = -(yv * at.softplus(-z) + (1 - yv) * at.softplus(z)).sum()
    xs, h0, U = at.fmatrix("xs"), at.fvector("h0"), at.fmatrix("U")
    hs, _ = ae.scan(lambda x_t, h, U: at.tanh(x_t + at.dot(h, U)), sequences=[xs],
                    outputs_info=[h0], non_sequences=[U])
    rng = np.random.default_rng(0)
    f32 = "float32"
    cases = [
        ([x, mu, sg], [at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum()],
         [rng.standard_normal((64, 48)), np.asarray(0.1), np.asarray(1.3)], [()]),
        ([A, B], [at.dot(A, B)], [rng.standard_normal((32, 16)).astype(f32),
                                  rng.standard_normal((16, 24)).astype(f32)], [(32, 24)]),
        ([X, w, b, yv], [logp] + ae.grad(logp, [w, b]),
         [rng.standard_normal((128, 64)).astype(f32), rng.standard_normal(64).astype(f32),
          np.asarray(0.1, f32), (rng.random(128) < 0.5).astype(f32)], [(), (64,), ()]),
        ([xs, h0, U], [hs], [rng.standard_normal((12, 64)).astype(f32), np.zeros(64, f32),
                             (rng.standard_normal((64, 64)) * 0.1).astype(f32)], [(12, 64)]),
    ]
    for fast in (True, False):
        for ins, outs, vals, shapes in cases:
            f = ae.function(ins, outs, mode=Mode(HipLinker(executor_factory=factory, fast_call=fast),
                                                 HIP_QUERY))
            f.trust_input = True
            for _ in range(2):
                res = f(*vals)
            res = res if isinstance(res, list) else [res]
            assert hasattr(f.vm, "slow_vm") == fast
            for r, shp, o in zip(res, shapes, outs):
                assert isinstance(r, DevArray) and r.shape == shp and r.dtype == o.dtype, (r, shp)
            trace = execs[-1].trace
            assert trace, "no C-ABI entry point was reached"
    # the Scan of the last case went through the persistent one-kernel loop
    assert list(execs[-1].scan_modes.values()) == ["persistent"], execs[-1].scan_modes
