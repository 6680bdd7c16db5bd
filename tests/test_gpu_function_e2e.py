"""The whole seam on hardware (SURVEY §8 H0 / boundary (b)): ``aesara.function(..., mode="HIP")``
of the REAL reference front end over the REAL ``PlanExecutor`` on an MI355X, in one process:

``FunctionMaker`` (compile/function/types.py:1586-1595, :1708) -> ``JITLinker.make_all``
(link/basic.py:684-747) -> ``HipLinker._fast_vm`` -> ``PlanExecutor`` -> C-ABI -> kernels, and back
through ``Function.__call__`` (types.py:791-1082): input filtering, ``trust_input``, device tensors
in / out, ``updates=`` into device-resident shared variables, fresh (non-recycled) outputs,
``allow_gc`` / ``free()``, a second shape signature, node-annotated errors.

The reference front end is not installed on the GPU box: it comes from the packed overlay
``oracle/_ref/aesara_ref_overlay.tar.gz`` (``oracle/pack_ref_overlay.sh``; git-ignored build
artefact that travels with the snapshot).  Skipped when neither that nor /root/reference exists.
The graphs are the golden cases' own builders (``oracle/gen_golden.py`` used as a library), the
expected values the committed outputs of the reference's ``Mode("cvm","fast_run")`` linker.
"""
import os

import numpy as np
import pytest

import ref_overlay
from golden_util import CASES, case_expected, case_inputs, assert_matches

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_overlay.available(),
                                 reason="no reference front end (oracle/_ref overlay not packed)")]

# default: EVERY golden case (271 graphs, ~2.5 min on the MI355X); AESARA_E2E_SUBSET=1 restricts
# the run to the BASELINE configs + one case per Op family
SUBSET = [
    "cfg1a_scalar_add", "cfg1b_matrix_add", "cfg2_gauss_sum", "cfg3a_gemv", "cfg3b_gemm_update",
    "cfg4_gru_b1_f32", "cfg4_gru_b8_f32", "cfg5_logistic",
    "ew_bcast8_float32", "ew_weird_strides", "ew_int_ops", "ew_bool_ops", "ew_transposed_3d",
    "dimshuffle3", "red4_f64", "red7_int", "red_fused_elemwise_axis", "red_bool_any_all",
    "gemm3_float64", "gemm_T110_float32", "gemm_empty045_float32", "dot_vec_combos",
    "gemv_T_beta_float64", "ger_f64", "batched_dot_f32", "blas_strides_float32",
    "subtensor_basic", "incsubtensor", "advsub1", "advincsub1_int", "advincsub1_f32", "alloc_join",
    "advsub_nd", "advincsub_mixed_slices", "subtensor_symbolic_edge_bounds",
    "scan_cumsum", "scan_taps", "scan_two_outputs", "scan_while_cumsum", "scan_grad_rnn",
    "scan_variant_3", "scan_nested_with_grad", "ifelse_lazy_c1",
    "softmax_family_float32", "softmax_grad_float64", "argmax_axes", "layernorm_float32",
    "nll_classifier_float32", "cumop_float", "mlp_layers_float32", "empty_inputs",
    "sp_gru_last_f32", "sp_lstm_vec_f32", "gru_bptt_b1_f32", "gru_bptt_b4_f32", "gru_bptt_b4_f64",
    "lstm_fused_fwd_f32", "lstm_fused_vec_bptt_f32", "rnn_bias_bptt_b4_f32", "sort_argsort",
]
BY_NAME = {c["name"]: c for c in CASES}
NAMES = SUBSET if os.environ.get("AESARA_E2E_SUBSET") else [c["name"] for c in CASES]


@pytest.fixture(scope="module")
def gg():
    """oracle/gen_golden.py as a library: the case builders (graph + input recipes)."""
    ref_overlay.import_reference()
    import gen_golden
    import aesara_amd
    aesara_amd.get_mode()            # registers linker "hip" / mode "HIP" with the reference
    return gen_golden


@pytest.fixture(scope="module")
def ae(gg):
    return gg.ae


def _builder(gg, name):
    return next(fn for n, fn, *_ in gg.CASES if n == name)


def _host(outs):
    import torch
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    return [o.detach().cpu().numpy() if isinstance(o, torch.Tensor) else np.asarray(o) for o in outs]


@pytest.mark.parametrize("name", NAMES)
def test_golden_case_through_aesara_function(gg, ae, name):
    """mode="HIP" by NAME, real executor: (1) host inputs through Function's own filter,
    (2) the replayed second call, (3) device tensors as ordinary (untrusted) arguments,
    (4) the same with ``trust_input``."""
    import torch
    c = BY_NAME[name]
    ins, outs, _specs = _builder(gg, name)()
    f = ae.function(ins, outs, mode="HIP", on_unused_input="ignore", accept_inplace=True)
    from aesara_amd.executor import PlanExecutor
    ex = f.maker.linker.executor
    assert isinstance(ex, PlanExecutor) and not ex.dry_run          # the real thing, no checker
    xs = case_inputs(c)
    want = case_expected(c)
    got = f(*xs)
    if name.startswith("cfg") and name != "cfg1a_scalar_add":
        assert all(isinstance(o, torch.Tensor) and o.is_cuda
                   for o in (got if isinstance(got, list) else [got])), "outputs stay in HBM"
    assert_matches(c, _host(got), want, "call 1 (host inputs)")
    assert_matches(c, _host(f(*xs)), want, "call 2 (replay)")
    # (0-d index-like values stay 0-d host values; uint16/32/64 have no torch device container)
    dev = [x if (x.ndim == 0 and x.dtype.kind in "iub") or not x.flags.c_contiguous
           or x.dtype.name in ("uint16", "uint32", "uint64")
           else torch.from_numpy(np.array(x, order="C")).cuda() for x in xs]
    assert_matches(c, _host(f(*dev)), want, "call 3 (device inputs through Function's filter)")
    f.trust_input = True
    assert_matches(c, _host(f(*dev)), want, "call 4 (device inputs, trust_input)")
    ex.check()


def test_cfg2_second_shape_signature_and_fresh_outputs(ae, gg):
    """One compiled function, two shapes (a new replay signature each), results are caller-owned:
    a later call never overwrites an earlier result (no_recycling / fresh-unless-borrowed,
    link/basic.py:719-725, types.py:1010-1019)."""
    import torch
    import aesara.tensor as at
    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    f = ae.function([x, mu, sg], at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum(), mode="HIP")
    f.trust_input = True
    rng = np.random.default_rng(5)
    m, s = np.asarray(0.1), np.asarray(1.3)
    kept = []
    for shape in ((256, 192), (512, 640), (256, 192), (512, 640), (300, 7)):
        xv = rng.standard_normal(shape)
        xd = torch.from_numpy(xv).cuda()
        r = f(xd, m, s)
        want = np.exp(-(xv - 0.1) ** 2 / (2 * 1.3 ** 2)).sum()
        np.testing.assert_allclose(r.item(), want, rtol=1e-12)
        kept.append((r, want))
    for r, want in kept:                       # earlier results are still what they were
        np.testing.assert_allclose(r.item(), want, rtol=1e-12)
    assert len({r.data_ptr() for r, _ in kept}) == len(kept)
    # the reference's own linker on the same host agrees (live oracle next to the goldens)
    f_ref = ae.function([x, mu, sg], at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum())
    np.testing.assert_allclose(kept[-1][0].item(), f_ref(xv, 0.1, 1.3), rtol=1e-12)


def test_cfg3b_hip_shared_updates_100_steps(ae, gg):
    """check_blas.py:54-57: ``updates=[(C, 0.4*C + 0.8*dot(A, B))]`` with device-resident state —
    through the REAL ``Function.__call__`` update protocol (types.py:1060-1069)."""
    import torch
    import aesara.tensor as at
    from aesara_amd.sharedvar import hip_shared
    rng = np.random.default_rng(0)
    n = 192
    A0 = (rng.standard_normal((n, n)) * 0.1).astype("float32")
    B0 = (rng.standard_normal((n, n)) * 0.1).astype("float32")
    C0 = rng.standard_normal((n, n)).astype("float32")
    A, B, Cs = hip_shared(A0, "A"), hip_shared(B0, "B"), hip_shared(C0, "C")
    f = ae.function([], [], updates=[(Cs, np.float32(0.4) * Cs + np.float32(0.8) * at.dot(A, B))],
                    mode="HIP")
    assert [nd.op for nd in f.maker.linker.plan.nodes] == ["Gemm"]
    ab = A0.astype("float64") @ B0
    want = C0.astype("float64")
    for step in range(100):
        assert f() == []
        v = Cs.container.storage[0]
        assert isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32
        want = 0.4 * want + 0.8 * ab
        if step in (0, 1, 9, 99):
            np.testing.assert_allclose(v.cpu().numpy(), want, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(Cs.get_value(), want, rtol=2e-5, atol=1e-6)
    # state set from the host between calls is picked up (uploaded once, new replay binding)
    Cs.set_value(np.zeros((n, n), "float32"))
    f()
    np.testing.assert_allclose(Cs.get_value(), 0.8 * ab, rtol=2e-5, atol=1e-6)


def test_training_loop_new_batches_and_updates(ae, gg):
    """SGD on config 5's graph: new device batch every call (rebinding), weights as
    ``hip_shared`` with ``updates=``; every step equals the reference's own linker."""
    import torch
    import aesara.tensor as at
    from aesara_amd.sharedvar import hip_shared
    rng = np.random.default_rng(3)
    D = 64
    w0 = (rng.standard_normal(D) / 8).astype("float32")
    X, y = at.fmatrix("X"), at.fvector("y")

    def build(shared):
        w, b = shared(w0, "w"), shared(np.float32(0.1), "b")
        p = at.sigmoid(at.dot(X, w) + b)
        logp = (y * at.log(p) + (1 - y) * at.log(1 - p)).mean()
        gw, gb = ae.grad(logp, [w, b])
        lr = np.float32(0.5)
        return w, b, [logp], [(w, w + lr * gw), (b, b + lr * gb)]

    w, b, outs, ups = build(hip_shared)
    f = ae.function([X, y], outs, updates=ups, mode="HIP")
    wr, br, outs_r, ups_r = build(lambda v, n: ae.shared(v, n))
    f_ref = ae.function([X, y], outs_r, updates=ups_r)
    f.trust_input = True
    for step in range(12):
        n = 2048 if step % 3 else 1024                     # two shape signatures, new tensors
        Xv = rng.standard_normal((n, D)).astype("float32")
        yv = (rng.random(n) < 0.5).astype("float32")
        (lp,) = f(torch.from_numpy(Xv).cuda(), torch.from_numpy(yv).cuda())
        (lp_ref,) = f_ref(Xv, yv)
        np.testing.assert_allclose(lp.item(), lp_ref, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(w.get_value(), wr.get_value(), rtol=2e-4, atol=2e-6)
    assert isinstance(w.container.storage[0], torch.Tensor) and w.container.storage[0].is_cuda
    np.testing.assert_allclose(b.get_value(), br.get_value(), rtol=2e-4, atol=2e-6)


def test_free_allow_gc_copy_and_numpy_mode(ae, gg):
    """``Function.free`` (types.py:1084-1100), ``Function.copy`` (:558, re-links through
    ``Linker.clone``/``accept``), and a linker instance that returns NumPy arrays."""
    import torch
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY, HipLinker
    x, v = at.dmatrix("x"), at.dvector("v")
    out = [at.dot(x, v) + 1.0, (x * 2).sum(axis=0)]
    f = ae.function([x, v], out, mode="HIP")
    xv, vv = np.random.default_rng(1).standard_normal((40, 30)), np.arange(30.0)
    r1 = _host(f(xv, vv))
    f.free()
    r2 = _host(f(xv, vv))
    g = f.copy()
    r3 = _host(g(xv, vv))
    for r in (r1, r2, r3):
        np.testing.assert_allclose(r[0], xv @ vv + 1.0, rtol=1e-12)
        np.testing.assert_allclose(r[1], (xv * 2).sum(axis=0), rtol=1e-12)
    fn = ae.function([x, v], out, mode=Mode(HipLinker(return_numpy=True), HIP_QUERY))
    rn = fn(xv, vv)
    assert all(isinstance(o, np.ndarray) for o in rn)
    np.testing.assert_allclose(rn[0], xv @ vv + 1.0, rtol=1e-12)
    # device tensors are ordinary arguments of an UNTRUSTED function: ``Function.__call__`` filters
    # them through the input cell's type (types.py:853-863), which is ``devcell.DeviceFilterType``
    # here — no ``trust_input`` escape hatch
    assert f.trust_input is False
    rd = f(torch.from_numpy(xv).cuda(), torch.from_numpy(vv).cuda())
    np.testing.assert_allclose(_host(rd)[0], xv @ vv + 1.0, rtol=1e-12)
    np.testing.assert_allclose(_host(rd)[1], (xv * 2).sum(axis=0), rtol=1e-12)


def test_untrusted_function_filters_device_tensors(ae, gg):
    """The decisions of ``TensorType.filter`` (tensor/type.py:135-256) for device-tensor
    arguments of an untrusted ``Function`` (types.py:853-863): exact dtype passes untouched,
    a safe upcast is converted ON THE DEVICE, a downcast is refused unless
    ``allow_input_downcast``, ``strict`` inputs take no conversion, wrong rank is a TypeError
    with the reference's "Bad input argument" prefix; mixed host / device arguments and a
    device tensor passed twice (aliased inputs, types.py:898-938) work."""
    import torch
    import aesara.tensor as at
    from aesara.compile.io import In
    x, v = at.dmatrix("x"), at.dvector("v")
    f = ae.function([x, v], at.dot(x, v) + x.sum(axis=0), mode="HIP")
    assert f.trust_input is False
    rng = np.random.default_rng(11)
    xv, vv = rng.standard_normal((30, 30)), rng.standard_normal(30)
    want = xv @ vv + xv.sum(axis=0)
    xd, vd = torch.from_numpy(xv).cuda(), torch.from_numpy(vv).cuda()
    np.testing.assert_allclose(_host(f(xd, vd))[0], want, rtol=1e-12)
    np.testing.assert_allclose(_host(f(xd, vv))[0], want, rtol=1e-12)        # device + host
    np.testing.assert_allclose(_host(f(xv, vd))[0], want, rtol=1e-12)
    # float32 device tensor into a float64 input: exact upcast, done by the cast kernel
    x32 = torch.from_numpy(xv.astype("float32")).cuda()
    got = _host(f(x32, vd))[0]
    np.testing.assert_allclose(got, xv.astype("float32").astype("float64") @ vv
                               + xv.astype("float32").astype("float64").sum(axis=0), rtol=1e-12)
    # wrong rank: the reference's message
    with pytest.raises(TypeError, match="(?s)Bad input argument.*Wrong number of dimensions"):
        f(vd, vd)
    # float64 -> float32 input: refused (precision), accepted with allow_input_downcast
    x4 = at.fmatrix("x4")
    g = ae.function([x4], (x4 * 2).sum(), mode="HIP")
    with pytest.raises(TypeError, match="(?s)Bad input argument.*without risking loss of precision"):
        g(xd)
    g2 = ae.function([x4], (x4 * 2).sum(), mode="HIP", allow_input_downcast=True)
    np.testing.assert_allclose(_host(g2(xd))[0], (xv.astype("float32") * 2).sum(), rtol=1e-5)
    # strict input: no conversion at all
    g3 = ae.function([In(x4, strict=True)], (x4 * 2).sum(), mode="HIP")
    with pytest.raises(TypeError, match="(?s)Bad input argument.*expected a tensor with dtype=float32"):
        g3(xd)
    np.testing.assert_allclose(_host(g3(x32))[0], (xv.astype("float32") * 2).sum(), rtol=1e-5)
    # the same device tensor as two arguments (aliased inputs): nothing is destroyed in place
    a, b = at.dmatrix("a"), at.dmatrix("b")
    h = ae.function([a, b], [a + b, a * b], mode="HIP")
    r = _host(h(xd, xd))
    np.testing.assert_allclose(r[0], 2 * xv, rtol=1e-15)
    np.testing.assert_allclose(r[1], xv * xv, rtol=1e-15)
    np.testing.assert_array_equal(xd.cpu().numpy(), xv)
    # keyword arguments go through Container.__set__ (types.py:894 ``self[k] = arg``)
    np.testing.assert_allclose(_host(f(x=xd, v=vd))[0], want, rtol=1e-12)


def test_memoized_argument_objects_follow_their_values(ae, gg):
    """Calls with the same argument OBJECTS take the executor's memo (``ReplayMixin._memoize``: one
    lookup on the arguments' identities); the memo must notice everything a caller can do to
    those objects between calls: a host scalar changed in place, a device matrix overwritten in
    place (same pointer: the launches read what is there now), another tensor, a tensor that died
    and whose id / pointer were re-used, a second shape — each against NumPy."""
    import gc
    import torch
    import aesara.tensor as at
    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    f = ae.function([x, mu, sg], at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum(), mode="HIP")
    assert f.trust_input is False
    ex = f.vm.jit_fn
    rng = np.random.default_rng(17)
    xs = [rng.standard_normal((96, 80)) for _ in range(3)]
    xd = [torch.from_numpy(a).cuda() for a in xs]
    m, s = np.asarray(0.1), np.asarray(1.3)

    def want(a, m_, s_):
        return np.exp(-(a - m_) ** 2 / (2 * s_ ** 2)).sum()
    for rep in range(4):                                  # rotation: three memo entries
        for a, d in zip(xs, xd):
            np.testing.assert_allclose(f(d, m, s).item(), want(a, 0.1, 1.3), rtol=1e-12)
    assert len(ex._memo) == 3
    m[...] = 0.4                                          # host scalar changed in place
    np.testing.assert_allclose(f(xd[0], m, s).item(), want(xs[0], 0.4, 1.3), rtol=1e-12)
    np.testing.assert_allclose(f(xd[0], m, s).item(), want(xs[0], 0.4, 1.3), rtol=1e-12)
    np.testing.assert_allclose(f(xd[0], m, s).item(), want(xs[0], 0.4, 1.3), rtol=1e-12)
    xd[1].copy_(xd[2])                                    # device matrix overwritten in place
    np.testing.assert_allclose(f(xd[1], m, s).item(), want(xs[2], 0.4, 1.3), rtol=1e-12)
    # a tensor dies, the next one may get its id and its pointer
    for k in range(6):
        a = rng.standard_normal((96, 80))
        d = torch.from_numpy(a).cuda()
        for _ in range(3):
            np.testing.assert_allclose(f(d, m, s).item(), want(a, 0.4, 1.3), rtol=1e-12)
        del d
        gc.collect()
    a = rng.standard_normal((50, 33))                     # another shape: a new signature
    d = torch.from_numpy(a).cuda()
    for _ in range(3):
        np.testing.assert_allclose(f(d, m, s).item(), want(a, 0.4, 1.3), rtol=1e-12)
    # results stay caller-owned on the memo path too
    r1, r2 = f(xd[0], m, s), f(xd[0], m, s)
    assert r1.data_ptr() != r2.data_ptr()
    ex.check()


def test_errors_name_the_apply_node(ae, gg):
    """A failure inside the thunk is re-raised by ``Function.__call__`` through
    ``raise_with_op`` (link/utils.py:270, types.py:974-991) with the Apply node that caused it —
    in the call that caused it, also on the replay path."""
    import aesara.tensor as at
    x, i = at.dmatrix("x"), at.lvector("i")
    f = ae.function([x, i], at.exp(x)[i] * 2.0, mode="HIP")
    xv = np.random.default_rng(2).standard_normal((10, 6))
    good = np.array([0, 3, -1], "int64")
    np.testing.assert_allclose(_host(f(xv, good))[0], np.exp(xv)[good] * 2, rtol=1e-12)
    np.testing.assert_allclose(_host(f(xv, good))[0], np.exp(xv)[good] * 2, rtol=1e-12)
    for _ in range(2):                                            # eager-recorded and replayed
        with pytest.raises(IndexError) as ei:
            f(xv, np.array([0, 10, 1], "int64"))
        assert "AdvancedSubtensor1" in str(ei.value) and "Apply node that caused the error" in str(ei.value)
        np.testing.assert_allclose(_host(f(xv, good))[0], np.exp(xv)[good] * 2, rtol=1e-12)
    a, b = at.dmatrix("a"), at.dmatrix("b")
    g = ae.function([a, b], a + b, mode="HIP")
    with pytest.raises(ValueError) as ev:
        g(np.zeros((3, 4)), np.zeros((5, 4)))
    assert "Elemwise" in str(ev.value) and "Apply node that caused the error" in str(ev.value)


def test_profile_is_filled_per_apply_node(ae, gg):
    """``profile=True``: ``ProfileStats.apply_time`` / ``apply_callcount`` per Apply node from HIP
    events around the plan steps (the role of link/vm.py:389-405 ``update_profile``)."""
    import aesara.tensor as at
    x, v = at.dmatrix("x"), at.dvector("v")
    f = ae.function([x, v], [at.dot(x, v).sum(), at.exp(x).sum(axis=0)], mode="HIP", profile=True)
    xv, vv = np.random.default_rng(1).standard_normal((400, 300)), np.arange(300.0)
    for _ in range(4):
        f(xv, vv)
    prof = f.profile
    assert prof.fct_callcount >= 3
    assert prof.apply_time and all(t >= 0 for t in prof.apply_time.values())
    assert sum(prof.apply_time.values()) > 0
    topo = f.maker.fgraph.toposort()          # fused steps are booked on their result's node
    assert len(prof.apply_time) >= 2 and all(n in topo for (_fg, n) in prof.apply_time)


def test_scan_with_shared_outputs_and_do_while_runs_as_one_launch(ae, gg):
    """Scan classes the element-wise kernel added in round 4 (aesara_amd/scan_persist_ew.py), through
    the real front end: a recurrence that also updates SHARED variables every step
    (``n_shared_outs``, scan/op.py:1673; tests/scan/test_basic.py:841
    test_shared_arguments_with_updates) and a do-while — both one launch, values equal to the
    reference's own linker run live on the host."""
    import aesara.tensor as at
    from aesara.scan.utils import until
    from aesara_amd.sharedvar import hip_shared
    rng = np.random.default_rng(21)
    xv = rng.uniform(0.1, 0.9, 40)

    def build(shared_ctor, mode):
        x = at.dvector("x")
        cnt = shared_ctor(np.asarray(2.0), name="cnt")
        tot = shared_ctor(np.asarray(0.0), name="tot")

        def step(x_t, acc):
            return acc * 0.9 + x_t * cnt, {cnt: cnt + 0.5, tot: tot + x_t * x_t}
        res, upd = ae.scan(step, sequences=[x], outputs_info=[at.as_tensor_variable(np.float64(0.0))])
        f = ae.function([x], res, updates=upd, mode=mode)
        w, _ = ae.scan(lambda x_t, s: (s + x_t, until(s + x_t > 3.0)), sequences=[x],
                       outputs_info=[at.as_tensor_variable(np.float64(0.0))])
        g = ae.function([x], [w, w.shape[0]], mode=mode)
        return f, g, cnt, tot
    f_h, g_h, cnt_h, tot_h = build(hip_shared, "HIP")
    f_r, g_r, cnt_r, tot_r = build(ae.shared, None)
    for call in range(3):
        np.testing.assert_allclose(_host(f_h(xv))[0], f_r(xv), rtol=1e-13)
        np.testing.assert_allclose(cnt_h.get_value(), cnt_r.get_value(), rtol=1e-13)
        np.testing.assert_allclose(tot_h.get_value(), tot_r.get_value(), rtol=1e-13)
    wh, nh = _host(g_h(xv))
    wr, nr = g_r(xv)
    assert int(nh) == int(nr) and 1 < int(nr) < 40
    np.testing.assert_allclose(wh, wr, rtol=1e-13)
    for fn in (f_h, g_h):
        modes = fn.maker.linker.executor.scan_modes
        assert list(modes.values()) == ["persistent"], modes
