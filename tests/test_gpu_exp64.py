"""The hand-written float64 ``exp`` (``exp_tbl64``, aesara_amd/codegen.py) on the device, against
the reference's ``Exp.c_code`` (scalar/basic.py:3102: the C library's ``exp``) run through the
reference's own C linker, and against NumPy: the edges of the range (|x| >= 708 takes the rare
branch: clamp + ``v_ldexp_f64``), the denormal results (-745.13 .. -708.4), overflow / underflow,
signed zeros, infinities, NaN, denormal arguments, and 10^6 random arguments over the whole finite
range.  Tolerance: <= 2 ulp on normal results (measured 1.02), <= 2 quanta (2 * 2^-1074) on
denormal results (``ldexp`` rounds a rounded mantissa a second time), specials exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPECIALS = np.array(
    [0.0, -0.0, 1.0, -1.0, 708.0, -708.0, 707.9999999999999, -707.9999999999999, 708.0000000000001,
     709.0, 709.78, 709.782712893384, 709.7827128933841, 709.79, 710.0, 1000.0, 1e300,
     -708.3964185322641, -708.3964185322642, -709.0, -720.0, -744.0, -745.0, -745.13,
     -745.1332191019411, -745.1332191019412, -745.14, -746.0, -1000.0, -1e300,
     np.inf, -np.inf, np.nan, -np.nan,
     5e-324, -5e-324, 2.2250738585072014e-308, -2.2250738585072014e-308, 1e-310, 1e-20, -1e-20,
     np.log(2.0), -np.log(2.0), 0.5 * np.log(2.0) / 64, 88.72283905206835, -87.33654475055310],
    dtype="float64")


def _inputs():
    rng = np.random.default_rng(20260924)
    return np.concatenate([
        SPECIALS,
        rng.uniform(-745.2, 709.8, 1_000_000),
        rng.uniform(-746.0, -707.0, 50_000),          # the whole denormal ramp
        rng.uniform(707.0, 710.0, 50_000),            # up to and across the overflow threshold
        rng.standard_normal(100_000),
        rng.standard_normal(10_000) * 1e-300,
    ]).astype("float64")


def _exp_plan(nd=1, with_sum=False):
    from aesara_amd.plan import Node, Plan, Var
    prog = {"n_in": 1, "nodes": [{"op": "exp", "in": [["i", 0]], "dtype": "float64"}], "out": [["t", 0]]}
    vs = {0: Var(0, "float64", [None] * nd), 1: Var(1, "float64", [None] * nd)}
    nodes = [Node("Elemwise", [0], [1], {"scalar": prog})]
    outs = [1]
    if with_sum:
        vs[2] = Var(2, "float64", [])
        nodes.append(Node("CAReduce", [1], [2], {"scalar_op": "add", "axis": None, "acc_dtype": "float64"}))
        outs = [2]
    return Plan("exp64", vs, [0], outs, nodes)


def _ulps(got, want):
    """Distance in representable doubles (denormals: in quanta) between two finite arrays."""
    def key(a):
        i = a.view(np.int64).copy()
        neg = i < 0
        i[neg] = np.int64(-(2 ** 63)) - i[neg]        # monotone integer image of the doubles
        return i
    # (both keys are the monotone images of two nearby non-negative doubles: the int64 difference
    # is exact; going through float64 would quantise it to 1024)
    return np.abs(key(np.ascontiguousarray(got)) - key(np.ascontiguousarray(want)))


def _check(got, want, x, label):
    assert got.dtype == np.float64 and got.shape == want.shape
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan), label
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf]), (label, x[inf][got[inf] != want[inf]][:5])
    # exact zeros of the reference may only differ by the smallest denormal's rounding boundary
    fin = ~(nan | inf)
    d = _ulps(got[fin], want[fin])
    worst = int(np.argmax(d)) if d.size else 0
    assert d.size == 0 or int(d.max()) <= 2, (label, x[fin][worst], got[fin][worst], want[fin][worst], int(d.max()))
    assert not np.any(np.signbit(got[fin])), label          # exp is never negative / never -0.0


def _device_exp(x, nd=1):
    import torch
    from aesara_amd.executor import PlanExecutor
    xs = x if nd == 1 else x[: (x.size // 1000) * 1000].reshape(-1, 1000)
    (o,) = PlanExecutor(_exp_plan(nd))(torch.from_numpy(np.ascontiguousarray(xs)).cuda())
    return xs, o.cpu().numpy()


def test_exp_tbl64_is_the_kernel_under_test(monkeypatch):
    """The float64 ``exp`` kernels these tests launch really are generated with the table form."""
    import torch
    from aesara_amd import exec_common, exec_elemwise, knobs
    from aesara_amd.executor import PlanExecutor
    assert int(knobs.get("FASTEXP")) == 1
    seen = []
    real = exec_common.load_kernels

    def spy(src, names):
        seen.append(src)
        return real(src, names)
    monkeypatch.setattr(exec_common, "load_kernels", spy)       # _Kernels.get (flat / n-d kernels)
    monkeypatch.setattr(exec_elemwise, "load_kernels", spy)     # row chains, tiled forms
    exec_common._Kernels.cache.clear()                          # (kernels other tests already loaded)
    for nd, shape in ((1, (4099,)), (2, (37, 1000))):
        PlanExecutor(_exp_plan(nd))(torch.zeros(shape, dtype=torch.float64, device="cuda"))
    PlanExecutor(_exp_plan(1, with_sum=True))(torch.zeros(4099, dtype=torch.float64, device="cuda"))
    assert len(seen) >= 2 and all("exp_tbl64(" in s.split("// ---- cross-lane")[-1] for s in seen), len(seen)


@pytest.mark.parametrize("nd", [1, 2])
def test_exp64_specials_and_range_against_numpy(nd):
    x = _inputs()
    xs, got = _device_exp(x, nd)
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        want = np.exp(xs)
    _check(got.reshape(-1), want.reshape(-1), xs.reshape(-1), "numpy")
    if nd == 1:
        k = len(SPECIALS)
        s, g = SPECIALS, got[:k]
        assert g[0] == 1.0 and g[1] == 1.0                                  # exp(+-0) == 1
        assert np.isinf(g[s == np.inf]).all() and (g[s == -np.inf] == 0.0).all()
        assert np.isinf(g[(s > 709.7827128933841) & np.isfinite(s)]).all()  # overflow -> +inf
        assert (g[s < -745.14] == 0.0).all()                                # underflow -> +0
        assert np.isfinite(g[s == 709.782712893384]).all()
        assert (g[np.abs(s) <= 5e-324] == 1.0).all()                        # denormal arguments
        assert g[list(s).index(-745.13)] == 5e-324                          # the smallest denormal


def test_exp64_against_the_reference_c_linker():
    """Same arguments through the reference's ``aesara.function(..., mode=Mode('cvm', 'fast_run'))``
    (``Exp.c_code``: the C library's ``exp``) on this box's host."""
    import ref_overlay
    if not ref_overlay.available():
        pytest.skip("no reference front end (oracle/_ref overlay not packed)")
    ae = ref_overlay.import_reference()
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    x = _inputs()
    xv = at.dvector("x")
    f = ae.function([xv], at.exp(xv), mode=Mode("cvm", "fast_run"))
    assert any(type(n.op).__name__ == "Elemwise" for n in f.maker.fgraph.toposort())
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        want = f(x)
    _xs, got = _device_exp(x, 1)
    _check(got, want, x, "reference C linker")
    # and through the drop-in boundary itself
    import aesara_amd
    aesara_amd.get_mode()
    from aesara_amd.linker import HIP_QUERY, HipLinker
    g = ae.function([xv], at.exp(xv), mode=Mode(HipLinker(return_numpy="all"), HIP_QUERY))
    _check(np.asarray(g(x)), want, x, "aesara.function(mode=HIP)")


def test_fused_gauss_sum_with_extreme_arguments():
    """Config 2's shape class with arguments that leave the fast range: the reduction's float64
    accumulator sees the denormal / zero / one results of the rare branch."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-760.0, -700.0, 200_000), rng.uniform(-40, 3, 200_000),
                        np.array([-np.inf, -0.0, 0.0, -745.13, -746.0])])
    (o,) = PlanExecutor(_exp_plan(1, with_sum=True))(torch.from_numpy(x).cuda())
    import math
    want = math.fsum(np.exp(x))
    np.testing.assert_allclose(float(o.cpu().numpy()), want, rtol=1e-13)
