#!/usr/bin/env python
"""Generate the golden parity fixtures under tests/golden/ by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE ONLY — runs in the authoring container (needs /root/reference); the GPU
box only consumes the committed outputs.

For every case below this script
  1. builds the graph with the reference front end (cases mirror the reference's own tests:
     tests/tensor/test_elemwise.py TestBroadcast :212 / TestCAReduce :412 / TestDimShuffle :47,
     tests/tensor/test_blas.py TestGemm :104 / BaseGemv :1545 / test_batched_dot :2650,
     tests/tensor/test_subtensor.py, tests/scan/test_basic.py, plus BASELINE.json configs 1-5),
  2. evaluates it with the reference's default linker ``Mode("cvm", "fast_run")`` (C thunks),
  3. lowers it with the HIP linker's rewrite query to a Plan (aesara_amd.lower) and evaluates
     that plan with the NumPy oracle (oracle/interp.py) — generation FAILS if the oracle
     disagrees with the reference (this is how the oracle is pinned),
  4. writes tests/golden/cases.json (plans + input recipes + tolerances) and
     tests/golden/<case>.npz (the reference's outputs).

Usage:  python oracle/gen_golden.py [--only PATTERN]
"""
import argparse
import fnmatch
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_overlay  # noqa: E402

ae = ref_overlay.import_reference()
import aesara.tensor as at  # noqa: E402
from aesara.compile.mode import Mode  # noqa: E402
from aesara.tensor.type import TensorType  # noqa: E402

import interp  # noqa: E402
from golden_inputs import make_input  # noqa: E402

from aesara_amd.linker import HIP_QUERY, HipLinker  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_MODE = Mode("cvm", "fast_run")

CASES = []


REF_PY = set()  # cases evaluated with the reference's Python linker (see batched_dot)


def case(name, exact=False, rtol=None, atol=None, ref_py=False):
    def deco(fn):
        if ref_py:
            REF_PY.add(name)
        CASES.append((name, fn, exact, rtol, atol))
        return fn
    return deco


def T(dtype, shape, name=None):
    return TensorType(dtype, shape=tuple(None if s != 1 else 1 for s in shape))(name)


def N(shape, dtype="float64", seed=0, scale=1.0, shift=0.0, view=None):
    d = {"kind": "normal", "seed": seed, "shape": list(shape), "dtype": dtype, "scale": scale,
         "shift": shift}
    if view:
        d["view"] = view
    return d


def U(shape, dtype="float64", seed=0, low=0.0, high=1.0, view=None):
    d = {"kind": "uniform", "seed": seed, "shape": list(shape), "dtype": dtype, "low": low,
         "high": high}
    if view:
        d["view"] = view
    return d


def I(shape, dtype="int32", seed=0, low=-10, high=10):  # noqa: E743
    return {"kind": "randint", "seed": seed, "shape": list(shape), "dtype": dtype, "low": low,
            "high": high}


def B(shape, seed=0, p=0.5, dtype="bool"):
    return {"kind": "bernoulli", "seed": seed, "shape": list(shape), "dtype": dtype, "p": p}


def K(value, dtype, shape=()):
    return {"kind": "const", "shape": list(shape), "dtype": dtype, "value": value}


# ---------------------------------------------------------------------------------------
# Elemwise (TestBroadcast shapes)
# ---------------------------------------------------------------------------------------
BCAST = [((3, 5), (3, 5)), ((3, 5), (1, 5)), ((3, 5), (3, 1)), ((1, 5), (5, 1)),
         ((1, 1), (1, 1)), ((1000,), (1000,)), ((40, 40), (40, 40)),
         ((2, 3, 4, 5), (2, 3, 4, 5)), ((2, 3, 4, 5), (1, 3, 1, 5)),
         ((2, 3, 4, 5), (1, 1, 1, 1)), ((), ())]

for _i, (_xs, _ys) in enumerate(BCAST):
    for _dt in ("float64", "float32"):
        def _mk(xs=_xs, ys=_ys, dt=_dt):
            x, y = T(dt, xs, "x"), T(dt, ys, "y")
            return [x, y], [x + y, x * y - x], [U(xs, dt, 1), U(ys, dt, 2)]
        case(f"ew_bcast{_i}_{_dt}")(_mk)


@case("ew_weird_strides")
def _():
    x, y = at.dmatrix("x"), at.dmatrix("y")
    return [x, y], [x + y, at.exp(x) * y], [
        N((6, 5), view={"kind": "step", "step": 2}), N((6, 5), seed=3, view={"kind": "transpose"})]


# scalar/math.py special functions (tests/scalar/test_math.py, tests/tensor/test_math_scipy.py):
# Gamma / GammaLn / Psi / TriGamma (the reference's own AS 103 / AS 121 C bodies), Erfcx / Erfinv /
# Erfcinv, Bessel J0 J1 I0 I1
for _dt, _rt, _at in (("float64", 2e-12, 2e-12), ("float32", 2e-5, 2e-6)):
    def _mkSP(dt=_dt):
        p, u, w = T(dt, (40,), "p"), T(dt, (40,), "u"), T(dt, (40,), "w")
        return [p, u, w], [at.gamma(p), at.gammaln(p), at.psi(p), at.tri_gamma(p), at.gammaln(p + 30) - at.gammaln(p),
                           at.erfcx(w), at.erfinv(u), at.erfcinv(u + 1), at.j0(w), at.j1(w), at.i0(u * 5),
                           at.i1(u * 5), at.psi(w), at.exp(at.gammaln(p) - at.gammaln(p + 0.5)) * at.erfcx(p)], \
            [U((40,), dt, 1, 0.05, 9.0), U((40,), dt, 2, -0.98, 0.98), U((40,), dt, 3, -4.0, 12.0)]
    case(f"ew_special_functions_{_dt}", rtol=_rt, atol=_at)(_mkSP)


# transposed operands (DimShuffle views inside the graph and transposed input views): the
# LDS-tiled Elemwise kernel, both tile sizes, ragged edges, mixed classes
for _dt, _r, _c in (("float64", 70, 45), ("float64", 130, 97), ("float32", 64, 128),
                    ("float32", 33, 211)):
    def _mkT(dt=_dt, r=_r, c=_c):
        x, y, v = T(dt, (r, c), "x"), T(dt, (c, r), "y"), T(dt, (c,), "v")
        return [x, y, v], [x + y.T, at.exp(y.T * 0.25) * x - v, at.sqr(y.T)], \
            [N((r, c), dt, 1), N((c, r), dt, 2), N((c,), dt, 3)]
    case(f"ew_transposed_{_dt}_{_r}x{_c}")(_mkT)


@case("ew_transposed_input_views")
def _():
    x, y, z = at.dmatrix("x"), at.dmatrix("y"), at.dmatrix("z")
    return [x, y, z], [x + y, x * y - z], [
        U((96, 80), seed=0), U((96, 80), seed=1, view={"kind": "transpose"}),
        U((96, 80), seed=2, view={"kind": "step", "step": 2})]


@case("ew_transposed_3d")
def _():
    x, y, z, w = at.dtensor3("x"), at.dtensor3("y"), at.dtensor3("z"), at.dmatrix("w")
    return [x, y, z, w], [x * y.dimshuffle(0, 2, 1) + z.dimshuffle(1, 2, 0)
                          - w.dimshuffle("x", 1, 0)], [
        N((5, 40, 50), seed=1), N((5, 50, 40), seed=2), N((50, 5, 40), seed=3), N((50, 40), seed=4)]


@case("ew_transposed_integer", exact=True)
def _():
    a, b, m = at.bmatrix("a"), at.bmatrix("b"), T("bool", (60, 37), "m")
    i, j = at.lmatrix("i"), at.lmatrix("j")
    return [a, b, m, i, j], [at.gt(a.T, b) & m.T, a.T + b, i.T * j - i.T // 7], [
        I((60, 37), "int8", 1, -100, 100), I((37, 60), "int8", 2, -100, 100), B((60, 37), 3),
        I((40, 50), "int64", 4, -1000, 1000), I((50, 40), "int64", 5, -1000, 1000)]


for _dt, _tol in (("float64", 1e-12), ("float32", 1e-5)):
    def _mkTR(dt=_dt):
        x, y = T(dt, (150, 70), "x"), T(dt, (70, 150), "y")
        return [x, y], [(x * y.T).sum(), at.max(y.T - x), at.sum(at.sqr(x.T) + y)], \
            [N((150, 70), dt, 1), N((70, 150), dt, 2)]
    case(f"reduce_all_transposed_{_dt}", rtol=_tol, atol=_tol)(_mkTR)


@case("ew_same_inputs")
def _():
    x = at.dmatrix("x")
    return [x], [x + x, x * x + x], [N((7, 9))]


@case("ew_transcendental_f64", rtol=1e-12, atol=1e-12)
def _():
    x, y = at.dmatrix("x"), at.dmatrix("y")
    e = at.tanh(x) * at.sigmoid(y) + at.softplus(x - y) + at.log1p(at.exp(-abs(x))) + \
        at.sqrt(abs(y)) + at.sqr(x) / (1 + at.sqr(y)) + at.cos(x) * at.sin(y) + at.erf(x)
    return [x, y], [e], [N((33, 65), scale=2.0), N((33, 65), seed=5, scale=2.0)]


@case("ew_transcendental_f32", rtol=2e-5, atol=2e-5)
def _():
    x, y = at.fmatrix("x"), at.fmatrix("y")
    e = at.tanh(x) * at.sigmoid(y) + at.softplus(x - y) + at.log1p(at.exp(-abs(x))) + \
        at.sqrt(abs(y)) + at.sqr(x) / (1 + at.sqr(y)) + at.expm1(x * 0.1) + at.log(abs(y) + 1)
    return [x, y], [e], [N((33, 65), "float32", scale=2.0), N((33, 65), "float32", 5, 2.0)]


@case("ew_softplus_ranges", rtol=1e-12, atol=0)
def _():
    x = at.dvector("x")
    return [x], [at.softplus(x), at.sigmoid(x)], [U((4000,), low=-60.0, high=60.0)]


@case("ew_int_ops", exact=True)
def _():
    a, b = at.imatrix("a"), at.imatrix("b")
    bb = at.switch(at.eq(b, 0), 3, b)
    return [a, b], [a // bb, a % bb, a * a - b, a & b, a | b, a ^ b, ~a, abs(a), -a,
                    at.maximum(a, b), at.minimum(a, b), at.sgn(a)], \
        [I((17, 23), seed=1, low=-50, high=50), I((17, 23), seed=2, low=-7, high=8)]


@case("ew_int8_uint8", exact=True)
def _():
    a, b = at.bmatrix("a"), TensorType("uint8", shape=(None, None))("b")
    return [a, b], [a + a, a * a, b + b, b * b, a // 3, b % 7, at.cast(a, "int32") + b], \
        [I((9, 31), "int8", 1, -128, 128), I((9, 31), "uint8", 2, 0, 256)]


@case("ew_compare_switch_clip", exact=True)
def _():
    x, y = at.dmatrix("x"), at.dmatrix("y")
    return [x, y], [at.lt(x, y), at.ge(x, y), at.eq(x, x), at.neq(x, y),
                    at.switch(at.gt(x, 0), x, y), at.clip(x, -0.5, 0.5),
                    at.and_(at.lt(x, y), at.gt(x, 0)), at.cast(x * 10, "int32"),
                    at.isnan(x / x * 0 + x), at.floor(x * 3), at.ceil(x * 3), at.round(x * 3)], \
        [N((21, 13)), N((21, 13), seed=9)]


@case("ew_bool_ops", exact=True)
def _():
    a, b = TensorType("bool", shape=(None,))("a"), TensorType("bool", shape=(None,))("b")
    return [a, b], [a & b, a | b, a ^ b, ~a, at.cast(a, "int8") + at.cast(b, "int8")], \
        [B((100,), 1), B((100,), 2)]


@case("ew_mixed_dtypes", rtol=1e-6, atol=1e-6)
def _():
    a, x, z = at.bvector("a"), at.fvector("x"), at.dvector("z")
    return [a, x, z], [a * x, x + z, a / 3, at.true_div(a, a + 130), at.cast(z, "float32") * x], \
        [I((257,), "int8", 1, -100, 100), N((257,), "float32", 2), N((257,), seed=3)]


@case("ew_6d", rtol=1e-12, atol=1e-12)
def _():
    x = TensorType("float64", shape=(None,) * 6)("x")
    y = TensorType("float64", shape=(None, 1, None, 1, None, 1))("y")
    return [x, y], [x * y + 1], [N((2, 3, 2, 3, 2, 3)), N((2, 1, 2, 1, 2, 1), seed=4)]


# ---------------------------------------------------------------------------------------
# DimShuffle (TestDimShuffle patterns) — output forces a materialising copy
# ---------------------------------------------------------------------------------------
DS = [((2, 3), (1, "x", 0)), ((1, 2, 3), (1, 2)), ((1, 2, 1, 3), (1, 3)), ((2, 3, 4), (2, 1, 0)),
      ((2, 3, 4), ("x", 2, 1, 0, "x")), ((1, 4, 3, 2, 1), (3, 2, 1)), ((1, 1, 4), (1, 2)),
      ((1, 1, 1), ()), ((1,), ("x", "x"))]
for _i, (_xs, _pat) in enumerate(DS):
    def _mk(xs=_xs, pat=_pat):
        x = T("float64", xs, "x")
        return [x], [x.dimshuffle(*pat), x.dimshuffle(*pat) * 2.0], [N(xs, seed=7)]
    case(f"dimshuffle{_i}", exact=True)(_mk)


@case("dimshuffle_int_T", exact=True)
def _():
    x = at.imatrix("x")
    return [x], [x.T, x.T + 1], [I((37, 53), seed=3)]


# ---------------------------------------------------------------------------------------
# CAReduce (TestCAReduce cases)
# ---------------------------------------------------------------------------------------
RED = [((5, 6), None), ((5, 6), (0, 1)), ((5, 6), (0,)), ((5, 6), (1,)), ((5, 6), (-1,)),
       ((5, 6), (-2,)), ((5, 6), ()), ((2, 3, 4, 5), (0, 1, 3)), ((2, 3, 4, 5), (-2, -3)),
       ((5, 0), None), ((5, 0), (0,)), ((5, 0), (1,)), ((5, 0), ()), ((), None), ((), ())]
for _i, (_xs, _ax) in enumerate(RED):
    def _mk(xs=_xs, ax=_ax):
        x = T("float64", xs, "x")
        outs = [at.sum(x, axis=ax), at.prod(x, axis=ax)]
        if 0 not in xs:
            outs += [at.max(x, axis=ax), at.min(x, axis=ax)]
        return [x], outs, [U(xs, seed=11, low=0.5, high=1.5)]
    case(f"red{_i}_f64", rtol=1e-12, atol=1e-12)(_mk)

    def _mk32(xs=_xs, ax=_ax):
        x = T("float32", xs, "x")
        return [x], [at.sum(x, axis=ax)], [U(xs, "float32", 11, 0.5, 1.5)]
    case(f"red{_i}_f32", rtol=1e-6, atol=1e-6)(_mk32)

    def _mki(xs=_xs, ax=_ax):
        x = T("int32", xs, "x")
        b = T("bool", xs, "b")
        i8 = T("int8", xs, "i8")
        outs = [at.sum(x, axis=ax), at.sum(i8, axis=ax), at.all(b, axis=ax), at.any(b, axis=ax)]
        if 0 not in xs:
            outs += [at.max(x, axis=ax), at.min(i8, axis=ax)]
        return [x, b, i8], outs, [I(xs, seed=5, low=-100, high=100), B(xs, 6, 0.8),
                                  I(xs, "int8", 7, -128, 128)]
    case(f"red{_i}_int", exact=True)(_mki)


@case("red_large_f32", rtol=2e-6, atol=1e-4)
def _():
    x = at.fmatrix("x")
    return [x], [x.sum(axis=0), x.sum(axis=1), x.sum(), x.max(axis=0), x.min(axis=1)], \
        [N((1500, 300), "float32", seed=21)]


@case("red_large_3d", rtol=1e-11, atol=1e-11)
def _():
    x = at.dtensor3("x")
    return [x], [x.sum(axis=(0, 2)), x.sum(axis=1), x.sum(axis=(1, 2)), x.max(axis=(0, 1))], \
        [N((64, 3, 200), seed=22)]


@case("red_tall_skinny", rtol=1e-6, atol=1e-3)
def _():
    x = at.fmatrix("x")
    return [x], [x.sum(axis=0), x.mean(axis=0)], [N((20000, 8), "float32", seed=23)]


@case("red_bool_any_all", exact=True)
def _():
    # sparse Trues / sparse Falses: a lane whose partial is already decided must still take part
    # in the cross-lane fold (TestCAReduce any/all, tests/tensor/test_elemwise.py:565-575)
    x, y = T("bool", (300, 70), "x"), T("bool", (300, 70), "y")
    return [x, y], [at.any(x, axis=0), at.any(x, axis=1), at.any(x), at.all(y, axis=0),
                    at.all(y, axis=1), at.all(y), at.any(x & y, axis=1), x.sum(axis=0)], \
        [B((300, 70), 31, 0.004), B((300, 70), 32, 0.996)]


@case("prod_grad_with_zeros", rtol=1e-12, atol=1e-12)
def _():
    # Prod.L_op (tensor/math.py:2577): rows with one zero get the product of the others through
    # ProdWithoutZeros = CAReduce(MulWithoutZeros :2713); rows with two zeros get 0
    x = at.dmatrix("x")
    return [x], [ae.grad(x.prod(axis=1).sum(), x), ae.grad(x.prod(), x), ae.grad(x.prod(axis=0).sum(), x)], \
        [I((6, 5), "float64", 5, -2, 4)]


@case("red_nan_propagation", exact=True)
def _():
    x = at.dmatrix("x")
    y = at.switch(at.gt(x, 2.5), np.nan, x)
    return [x], [at.max(y, axis=0), at.min(y, axis=1), at.max(y)], [N((40, 30), seed=24)]


@case("red_fused_elemwise_axis", rtol=1e-12, atol=1e-12)
def _():
    x, y = at.dmatrix("x"), at.dvector("y")
    e = at.exp(-(x - y) ** 2)
    return [x, y], [e.sum(axis=0), e.sum(axis=1), (e * 2).sum(), e], \
        [N((50, 70), seed=25), N((70,), seed=26)]


# ---------------------------------------------------------------------------------------
# BLAS (TestGemm.cmp / BaseGemv / TestBlasStrides / test_batched_dot)
# ---------------------------------------------------------------------------------------
def _gemm_case(name, dt, M, K, Nn, a, b, tx=False, ty=False, tz=False, rtol=None):
    def mk():
        z, x, y = (T(dt, (2, 2), n) for n in "zxy")
        zs = N((M, Nn), dt, 1, view={"kind": "transpose"} if tz else None)
        xs = N((M, K), dt, 2, view={"kind": "transpose"} if tx else None)
        ys = N((K, Nn), dt, 3, view={"kind": "transpose"} if ty else None)
        return [z, x, y], [b * z + a * at.dot(x, y)], [zs, xs, ys]
    tol = rtol or (1e-12 if dt == "float64" else 2e-5)
    case(name, rtol=tol, atol=tol * 10)(mk)


_k = 0
for _dt in ("float64", "float32"):
    for (_M, _K, _N) in [(3, 4, 5), (4, 5, 1), (1, 7, 1), (130, 70, 250), (256, 256, 256)]:
        for (_a, _b) in [(1.0, 0.0), (1.0, 1.0), (-1.0, 0.6), (0.6, -1.0), (0.0, 1.0)]:
            if (_M, _K, _N) != (3, 4, 5) and (_a, _b) not in [(1.0, 0.0), (0.6, -1.0)]:
                continue
            _gemm_case(f"gemm{_k}_{_dt}", _dt, _M, _K, _N, _a, _b)
            _k += 1
    for (_tx, _ty, _tz) in [(1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 1, 1)]:
        _gemm_case(f"gemm_T{_tx}{_ty}{_tz}_{_dt}", _dt, 132, 68, 200, 0.8, 0.4, _tx, _ty, _tz)
    for (_M, _K, _N) in [(0, 4, 5), (3, 0, 5), (3, 4, 0)]:
        _gemm_case(f"gemm_empty{_M}{_K}{_N}_{_dt}", _dt, _M, _K, _N, 0.5, 2.0)


@case("dot22_f32", rtol=2e-5, atol=2e-4)
def _():
    x, y = at.fmatrix("x"), at.fmatrix("y")
    return [x, y], [at.dot(x, y), at.dot(x, y) * np.float32(0.25)], \
        [N((77, 129), "float32", 1), N((129, 65), "float32", 2)]


@case("dot_vec_combos", rtol=1e-12, atol=1e-11)
def _():
    A, v, w = at.dmatrix("A"), at.dvector("v"), at.dvector("w")
    return [A, v, w], [at.dot(A, v), at.dot(w, A), at.dot(w, at.dot(A, v))], \
        [N((40, 60), seed=1), N((60,), seed=2), N((40,), seed=3)]


# independent outputs of ONE scalar program over operands of different shapes (towers of a model,
# per-parameter norms): the executor sends them out as one horizontally fused launch
@case("hfuse_towers_f64", rtol=1e-12, atol=1e-12)
def _():
    xs = [at.dmatrix("x%d" % k) for k in range(5)]
    return xs, [at.exp(-0.5 * x ** 2).sum() for x in xs], \
        [N((64, 32), seed=1), N((200, 120), seed=2), N((8, 4), seed=3), N((333, 64), seed=4),
         N((96, 1000), seed=5)]


@case("hfuse_norms_f32", rtol=2e-5, atol=1e-6)
def _():
    W, b, V, c = at.fmatrix("W"), at.fvector("b"), at.fmatrix("V"), at.fvector("c")
    ps = [W, b, V, c]
    sq = [at.sqr(p).sum() for p in ps]
    return ps, sq + [at.sqrt(sq[0] + sq[1] + sq[2] + sq[3]), abs(W).max(), abs(V).max()], \
        [N((300, 200), "float32", 1), N((200,), "float32", 2), N((200, 48), "float32", 3),
         N((48,), "float32", 4)]


# integer / bool / mixed-dtype Dot (tensor/math.py:1879; tests/tensor/test_math.py TestDot): NumPy's
# wrap-around arithmetic, bit for bit; a mixed pair is computed in the common type
for _dt, _lo, _hi in (("int8", -128, 127), ("int16", -3000, 3000), ("int32", -70000, 70000),
                      ("int64", -(2 ** 40), 2 ** 40), ("uint8", 0, 255), ("uint32", 0, 2 ** 31)):
    def _mkID(dt=_dt, lo=_lo, hi=_hi):
        A, Bm, v, w = T(dt, (2, 2), "A"), T(dt, (2, 2), "B"), T(dt, (2,), "v"), T(dt, (2,), "w")
        return [A, Bm, v, w], [at.dot(A, Bm), at.dot(A, v), at.dot(w, A), at.dot(w, at.dot(A, v)),
                               at.dot(Bm.T[::2], A.T[:, ::-1])], \
            [I((70, 45), dt, 1, lo, hi), I((45, 83), dt, 2, lo, hi), I((45,), dt, 3, lo, hi),
             I((70,), dt, 4, lo, hi)]
    case(f"dot_{_dt}", exact=True)(_mkID)


@case("dot_bool", exact=True)
def _():
    A, Bm, v = T("bool", (2, 2), "A"), T("bool", (2, 2), "B"), T("bool", (2,), "v")
    return [A, Bm, v], [at.dot(A, Bm), at.dot(A, v)], [B((33, 70), 1, 0.05), B((70, 20), 2, 0.05),
                                                      B((70,), 3, 0.05)]


@case("dot_mixed", rtol=1e-12, atol=1e-9)
def _():
    Ai, Bf, Cd, Dl, v8 = at.imatrix("Ai"), at.fmatrix("Bf"), at.dmatrix("Cd"), at.lmatrix("Dl"), \
        at.bvector("v8")
    return [Ai, Bf, Cd, Dl, v8], [at.dot(Ai, Bf), at.dot(Bf.T, Cd), at.dot(Ai, Dl), at.dot(Dl.T, v8),
                                  at.dot(Cd.T, Dl)], \
        [I((30, 40), "int32", 1, -1000, 1000), N((40, 30), "float32", 2), N((40, 25), "float64", 3),
         I((40, 30), "int64", 4, -(2 ** 36), 2 ** 36), I((40,), "int8", 5, -100, 100)]


@case("batched_dot_int32", exact=True, ref_py=True)
def _():
    x, y = at.itensor3("x"), at.itensor3("y")
    return [x, y], [at.batched_dot(x, y)], [I((5, 17, 40), "int32", 1, -70000, 70000),
                                            I((5, 40, 9), "int32", 2, -70000, 70000)]


def _gemv_case(name, dt, M, Nn, alpha, beta, tA=False, rtol=None):
    def mk():
        y, A, x = T(dt, (2,), "y"), T(dt, (2, 2), "A"), T(dt, (2,), "x")
        return [y, A, x], [beta * y + alpha * at.dot(A, x)], \
            [N((M,), dt, 1), N((M, Nn), dt, 2, view={"kind": "transpose"} if tA else None),
             N((Nn,), dt, 3)]
    tol = rtol or (1e-12 if dt == "float64" else 3e-5)
    case(name, rtol=tol, atol=tol * 10)(mk)


for _dt in ("float64", "float32"):
    _gemv_case(f"gemv_small_{_dt}", _dt, 3, 5, 1.0, 0.0)
    _gemv_case(f"gemv_beta_{_dt}", _dt, 200, 300, 0.5, -0.3)
    _gemv_case(f"gemv_T_{_dt}", _dt, 256, 3000, 1.0, 0.0, tA=True)
    _gemv_case(f"gemv_T_beta_{_dt}", _dt, 30, 50, 2.0, 1.0, tA=True)
    _gemv_case(f"gemv_wide_{_dt}", _dt, 7, 4096, 1.0, 0.0)
    _gemv_case(f"gemv_empty_{_dt}", _dt, 5, 0, 1.0, 1.0)


@case("ger_f64", rtol=1e-12, atol=1e-12)
def _():
    A, x, y = at.dmatrix("A"), at.dvector("x"), at.dvector("y")
    return [A, x, y], [A + 0.3 * at.outer(x, y)], [N((30, 40), seed=1), N((30,), seed=2),
                                                   N((40,), seed=3)]


# BatchedDot's C thunk references sgemm_/dgemm_ directly and cannot link without blas__ldflags
# (tensor/blas.py:2241); the reference's own Python linker (perform :2224) is the oracle here.
@case("batched_dot_f32", rtol=2e-5, atol=2e-4, ref_py=True)
def _():
    x, y = at.ftensor3("x"), at.ftensor3("y")
    return [x, y], [at.batched_dot(x, y)], [N((7, 33, 40), "float32", 1),
                                            N((7, 40, 21), "float32", 2)]


@case("batched_dot_f64", rtol=1e-12, atol=1e-11, ref_py=True)
def _():
    x, y = at.dtensor3("x"), at.dtensor3("y")
    return [x, y], [at.batched_dot(x, y)], [N((3, 130, 17), seed=1), N((3, 17, 140), seed=2)]


# ---------------------------------------------------------------------------------------
# Subtensor / IncSubtensor / AdvancedSubtensor1 / Alloc / Join (bit-exact)
# ---------------------------------------------------------------------------------------
# IfElse (tests/test_ifelse.py: test_lazy_if :79, test_multiple_out :176, test_grad_lazy_if :115,
# test_nested / test_pushout*): scalar condition, several outputs, gradients, nesting, branches with
# work the lazy VM skips
for _cv in (0, 1):
    @case(f"ifelse_lazy_c{_cv}", rtol=1e-12, atol=1e-12)
    def _(cv=_cv):
        from aesara.ifelse import ifelse
        x, y, c = at.dvector("x"), at.dvector("y"), at.iscalar("c")
        z = ifelse(c, at.exp(x) * 2, at.dot(at.outer(y, y), y) - 1)         # expensive else-branch
        z1, z2 = ifelse(c, (x + 1, y * 2), (x * x, y - 3))                     # test_multiple_out
        g = ae.grad(ifelse(c, (x ** 2).sum(), (y ** 3).sum()), [x, y])       # test_grad_lazy_if
        inner = ifelse(at.gt(x.sum(), 0), x.max(), y.min())                  # condition computed on device
        nested = ifelse(c, ifelse(at.lt(y[0], 0), x * 3, x * 5), ifelse(c, x, x - y))
        return [x, y, c], [z, z1, z2, g[0], g[1], inner + 0, nested], \
            [N((7,), seed=1), N((7,), seed=2), K(cv, "int32")]


@case("ifelse_in_scan_and_shapes", rtol=1e-12, atol=1e-12)
def _():
    from aesara.ifelse import ifelse
    x, c = at.dmatrix("x"), at.iscalar("c")
    # branches of different run-time shapes (IfElse only requires equal types)
    sh = ifelse(c, x[:2], x[1:])
    # an IfElse inside a Scan step: accumulate rows with a data-dependent rule
    acc, _ = ae.scan(lambda row, a: ifelse(at.gt(row.sum(), 0), a + row, a - 2 * row),
                     sequences=[x], outputs_info=[at.zeros_like(x[0])])
    return [x, c], [sh * 2, acc[-1], ifelse(at.eq(c, 0), x.sum(axis=0), x.sum(axis=0) * 0)], \
        [N((5, 4), seed=3), K(0, "int32")]


# TestBlasStrides (tests/tensor/test_blas.py:2085-2500): Dot22 / Gemm / Gemv / Ger on operands that
# are stepped, reversed and transposed views of larger buffers
for _dt, _tol in (("float64", 1e-12), ("float32", 2e-5)):
    def _mkBS(dt=_dt):
        a, b, c = T(dt, (8, 12), "a"), T(dt, (12, 10), "b"), T(dt, (8, 10), "c")
        v, w = T(dt, (12,), "v"), T(dt, (8,), "w")
        k4 = np.asarray(0.4, dt)
        k8 = np.asarray(0.8, dt)
        outs = [at.dot(a[::2, ::3], b[::3, ::2]), at.dot(a[::-1, ::-3], b[::-3, ::-1]),
                at.dot(a.T[::3, ::2].T, b[::3]), at.dot(b.T[::2, ::3], a.T[::3, ::-2]),
                k4 * c[::2, ::2] + k8 * at.dot(a[::2, ::3], b[::3, ::2]),
                k4 * c[::-2, ::-2] + k8 * at.dot(a[::-2, ::3], b[::3, ::-2]),
                at.dot(a[::2, ::-1], v[::-1]), w[::2] * k4 + k8 * at.dot(a[::2], v),
                at.dot(a.T[::-2, ::2], w[::2]), at.dot(v[::3], b[::3, ::-1]),
                c[::2, ::5] + k4 * at.outer(w[::-2], v[::6]), at.dot(v[::-2], v[::2])]
        return [a, b, c, v, w], outs, [N((8, 12), dt, 1), N((12, 10), dt, 2), N((8, 10), dt, 3),
                                       N((12,), dt, 4), N((8,), dt, 5)]
    case(f"blas_strides_{_dt}", rtol=_tol, atol=_tol)(_mkBS)


@case("gemv_symbolic_alpha_beta", rtol=1e-12, atol=1e-12)
def _():
    # alpha / beta of Gemv as run-time 0-d values, row-major and transposed matrices
    A, v, w, s, t = at.dmatrix("A"), at.dvector("v"), at.dvector("w"), at.dscalar("s"), at.dscalar("t")
    return [A, v, w, s, t], [s * at.dot(A, v) + w, t * w + s * at.dot(A, v), s * at.dot(A.T, w) + t * v,
                             at.dot(A.T[::2], w) * s], \
        [N((9, 14), seed=1), N((14,), seed=2), N((9,), seed=3), K(0.7, "float64"), K(-1.3, "float64")]


@case("subtensor_basic", exact=True)
def _():
    x = at.imatrix("x")
    i = at.lscalar("i")
    return [x, i], [x[1:5], x[::-1], x[2], x[1:, ::2], x[i], x[-i:, 1], x[::-2, ::-1] + 0], \
        [I((9, 11), seed=1), K(3, "int64")]


# modelled on tests/tensor/test_subtensor.py (TestSubtensor.test_slice_*, test_ellipsis, test_newaxis,
# test_noncontiguous, test_shape_i_*): 3-d views, out-of-range slice bounds (clipped like NumPy),
# negative steps, symbolic bounds / steps, new axes and ellipsis — all bit-exact index work
@case("subtensor_3d_slices", exact=True)
def _():
    x = at.itensor3("x")
    a, b, st = at.lscalar("a"), at.lscalar("b"), at.lscalar("st")
    outs = [x[1:, :, ::2], x[-3:-1], x[:, -1], x[..., 2], x[2, ...], x[1, 2, 3:], x[::-1, ::-2, ::-3],
            x[100:], x[:100], x[-100:2], x[5:2], x[2:5:-1], x[::5], x[:, None, 1:3], x[None, 0, :, None],
            x[a:b], x[a:b:st], x[::st], x[b:a:-st], x[-a:], x[:, a, b:], x[a], x[-a, -b, -a]]
    return [x, a, b, st], [o + 0 for o in outs], [I((6, 7, 8), seed=1), K(1, "int64"), K(5, "int64"),
                                                   K(2, "int64")]


@case("subtensor_symbolic_edge_bounds", exact=True)
def _():
    v = at.lvector("v")
    a, b, st = at.lscalar("a"), at.lscalar("b"), at.lscalar("st")
    outs = [v[a:b:st], v[b:a:st], v[a::st], v[:b:st], v[a:b], v[-b:-a], v[::st], v[st], v[-st]]
    return [v, a, b, st], [o * 1 for o in outs], [I((17,), "int64", seed=2, low=-99, high=99),
                                                   K(-20, "int64"), K(30, "int64"), K(-3, "int64")]


@case("subtensor_of_views_and_dimshuffles", exact=True)
def _():
    x = at.lmatrix("x")
    y = x.T[1:, ::2]                       # view of a view of a transposed view
    z = x.dimshuffle(1, "x", 0)[2:, :, ::-1]
    return [x], [y + 0, y[::-1, 1:] * 2, z + 0, z[0, 0, 1:4] - 1, x[::2][1:][::-1] + 0], \
        [I((9, 12), "int64", seed=3, low=-1000, high=1000)]


@case("incsubtensor_views_steps_broadcast", exact=True)
def _():
    x, y, r = at.ltensor3("x"), at.lmatrix("y"), at.lvector("r")
    s = at.lscalar("s")
    return [x, y, r, s], [
        at.set_subtensor(x[1], y), at.inc_subtensor(x[:, 2], y[:4, :5]), at.inc_subtensor(x[::-1, ::2, 1], 3),
        at.set_subtensor(x[..., -1], r[:6]), at.inc_subtensor(x[1:3, 1:3, 1:3], s),
        at.set_subtensor(x[::2, ::-3, ::2], y[None, :2, :3]), at.inc_subtensor(x[2:, -2:, :][0], y[:2, :5] * 0 + r[:5]),
        at.inc_subtensor(at.set_subtensor(x[0], 1)[0, 1], r[:5])], \
        [I((4, 6, 5), "int64", seed=1), I((6, 5), "int64", seed=2), I((8,), "int64", seed=3), K(11, "int64")]


@case("advsub1_variants", exact=True, ref_py=True)
def _():
    x, m = at.ltensor3("x"), at.lmatrix("m")
    idx, idx8, j = at.lvector("idx"), at.bvector("idx8"), at.bvector("j")
    return [x, m, idx, idx8, j], [x[idx], m[idx8], m.T[idx8], x[idx][:, 1], m[idx][::-1], x[:, j], m[:, idx8], x[..., j],
                               at.inc_subtensor(m[idx8], 5), at.set_subtensor(m[idx[:3]], m[:3] * 2)], \
        [I((7, 3, 4), "int64", seed=1), I((7, 5), "int64", seed=2), I((11,), "int64", 3, -7, 7),
         I((4,), "int8", 4, -5, 5), I((3,), "int8", 5, -3, 3)]


# integer arrays mixed with slices / newaxis (NumPy placement rules: adjacent arrays keep their
# position, separated ones move the broadcast block to the front) — tests/tensor/test_subtensor.py
# TestAdvancedSubtensor.test_adv_subtensor_w_slice / test_advinc_subtensor / test_adv_sub_3d
@case("advsub_mixed_slices", exact=True, ref_py=True)
def _():
    x = at.ltensor4("x")
    i, j, m = at.lvector("i"), at.lvector("j"), at.lmatrix("m")
    a = at.lscalar("a")
    outs = [x[i, 1:3], x[1:, i], x[:, i, j], x[i, :, j], x[:, :, i, j], x[::2, m, 1:], x[i, None, j],
            x[a:, i, ::-1], x[:, m[:, :2], :, i[:2]], x[..., i], x[i, ..., j]]
    return [x, i, j, m, a], outs, [I((5, 4, 6, 3), "int64", seed=1, low=-99, high=99),
                                   I((3,), "int64", 2, -3, 3), I((3,), "int64", 3, -3, 3),
                                   I((2, 3), "int64", 4, -3, 3), K(1, "int64")]


@case("advincsub_mixed_slices", exact=True, ref_py=True)
def _():
    x, y = at.ltensor3("x"), at.lmatrix("y")
    i, j = at.lvector("i"), at.lvector("j")
    return [x, y, i, j], [at.inc_subtensor(x[i, 1:3], 5), at.set_subtensor(x[:, i], y[:3, :4][None] * 0 + 7),
                          at.inc_subtensor(x[:, i, j], y[:4, :3]), at.set_subtensor(x[i, :, j], y[:3, :5]),
                          at.inc_subtensor(x[1:, i, ::2], y[:3, :2]), at.inc_subtensor(x[i, ::-1, j[0]], y[:3, :5])], \
        [I((4, 5, 4), "int64", seed=1), I((6, 6), "int64", seed=2), I((3,), "int64", 3, -4, 4),
         I((3,), "int64", 4, -4, 4)]


# MatMul (tests/tensor/test_math.py TestMatMul :3440): stacks, broadcast batch dims, promoted vectors
for _dt, _tol in (("float64", 1e-12), ("float32", 2e-5)):
    def _mkMM(dt=_dt):
        # (the reference's MatMul.make_node only accepts fully static N-d shapes: math.py:2924)
        def TS(shp, name):
            return TensorType(dt, shape=tuple(shp))(name)
        a3, b3, m, v = TS((4, 5, 6), "a3"), TS((4, 6, 3), "b3"), TS((6, 3), "m"), TS((6,), "v")
        a4, b4 = TS((2, 1, 5, 6), "a4"), TS((1, 3, 6, 2), "b4")
        mm = lambda p, q: at.matmul(p, q, dtype=dt)  # noqa: E731  (default: config.floatX)
        return [a3, b3, m, v, a4, b4], [mm(a3, b3), mm(a3, m), mm(a3, v), mm(v, b3),
                                        mm(v, v), mm(a4, b4), mm(m.T, m),
                                        mm(b3.dimshuffle(0, 2, 1), a3.dimshuffle(0, 2, 1)),
                                        mm(a4, m), mm(b4.dimshuffle(0, 1, 3, 2), a4.dimshuffle(0, 1, 3, 2))], \
            [N((4, 5, 6), dt, 1), N((4, 6, 3), dt, 2), N((6, 3), dt, 3), N((6,), dt, 4),
             N((2, 1, 5, 6), dt, 5), N((1, 3, 6, 2), dt, 6)]
    case(f"matmul_{_dt}", rtol=_tol, atol=_tol, ref_py=True)(_mkMM)


# Eye / Tri / ExtractDiag / AllocDiag (tests/tensor/test_basic.py TestEye :841, TestTriangle :905,
# TestDiag / test_diag* :3593-3720)
@case("eye_tri_diag", exact=True)
def _():
    n, m = at.lscalar("n"), at.lscalar("m")
    x, t3, v = at.lmatrix("x"), at.ltensor3("t3"), at.lvector("v")
    return [n, m, x, t3, v], [
        at.eye(n, m, 0, dtype="int64") * 3, at.eye(n, m, 2, dtype="int32"), at.eye(m, n, -3, dtype="int8"),
        at.eye(n, n, 9, dtype="int64"), at.tri(n, m, 0, dtype="int64"), at.tri(n, m, 2, dtype="int32"),
        at.tri(m, n, -2, dtype="int8"), at.tril(x), at.triu(x, 1), at.tril(x.T, -1),
        at.diag(x), at.diagonal(x, offset=2), at.diagonal(x, offset=-3), at.diagonal(x.T[::2], offset=1),
        at.diagonal(t3, offset=1, axis1=0, axis2=2), at.diagonal(t3, offset=-1, axis1=2, axis2=1),
        at.diag(v), at.diag(v, 2), at.diag(v, -1), at.diag(at.diag(x[:5, :5]) * 2),
        at.basic.AllocDiag(offset=1, axis1=0, axis2=2)(x[:3, :4]),
        at.extra_ops.fill_diagonal(x, 7), at.extra_ops.fill_diagonal(x.T, v[0]),
        at.extra_ops.fill_diagonal(t3[:, :4, :4], -1)], \
        [K(5, "int64"), K(7, "int64"), I((6, 8), "int64", seed=1), I((4, 5, 6), "int64", seed=2),
         I((4,), "int64", seed=3)]


# an end-to-end probabilistic-model graph: hierarchical normal + Poisson log-density over a packed
# parameter vector with data embedded as graph constants, and its gradient (what a sampler or
# optimiser evaluates per step): Subtensor / AdvancedSubtensor1 + scatter-add in the gradient,
# Gemv, fused Elemwise / Sum chains, gammaln, switch bounds
@case("hierarchical_logp_and_grad", rtol=1e-11, atol=1e-11, ref_py=True)
def _():
    rng = np.random.default_rng(0)
    Nn, G, D = 500, 8, 3
    Xd, gidx, yd = rng.standard_normal((Nn, D)), rng.integers(0, G, Nn), rng.standard_normal(Nn)
    cnt = rng.poisson(3.0, Nn)
    theta = at.dvector("theta")
    mu_g, log_sd_g, beta = theta[:G], theta[G], theta[G + 1:G + 1 + D]
    log_sigma, log_lam = theta[G + 1 + D], theta[G + 2 + D]
    sd_g, sigma = at.exp(log_sd_g), at.exp(log_sigma)

    def normal_logp(v, m, sd):
        return -0.5 * ((v - m) / sd) ** 2 - at.log(sd) - 0.5 * np.log(2 * np.pi)
    lp = normal_logp(mu_g, 0.0, sd_g).sum() + normal_logp(beta, 0.0, 10.0).sum() + normal_logp(log_sd_g, 0.0, 1.0)
    mu = mu_g[gidx] + at.dot(at.as_tensor(Xd), beta)
    lp += normal_logp(at.as_tensor(yd), mu, sigma).sum()
    lam = at.exp(log_lam + 0.1 * mu)
    lp += (at.as_tensor(cnt) * at.log(lam) - lam - at.gammaln(at.as_tensor(cnt) + 1.0)).sum()
    lp += at.switch(sigma > 50, -np.inf, 0.0) + log_sd_g + log_sigma
    return [theta], [lp, ae.grad(lp, theta)], [N((G + 3 + D,), seed=5, scale=0.3)]


# RNN language-model step: embedding gather -> Scan (tanh RNN over a batch of states) -> nnet
# softmax cross-entropy -> gradients w.r.t. all six parameters (scatter-add into the embedding,
# BPTT through the Scan)
@case("rnn_lm_loss_and_grads", rtol=1e-10, atol=1e-11, ref_py=True)
def _():
    import aesara.tensor.nnet as nn
    V, E, H, Tt, Bb = 30, 8, 12, 7, 5
    shapes = ((V, E), (E, H), (H, H), (H,), (H, V), (V,))
    params = [T("float64", sh, f"p{k}") for k, sh in enumerate(shapes)]
    Wemb, Wx, Wh, bh, Wo, bo = params
    tok, tgt = at.lmatrix("tok"), at.lmatrix("tgt")
    emb = Wemb[tok.flatten()].reshape((Tt, Bb, E))
    hs, _ = ae.scan(lambda x_t, h: at.tanh(at.dot(x_t, Wx) + at.dot(h, Wh) + bh),
                    sequences=[emb], outputs_info=[at.zeros((Bb, H))])
    logits = at.dot(hs.reshape((Tt * Bb, H)), Wo) + bo
    loss = nn.categorical_crossentropy(nn.softmax(logits), tgt.flatten()).mean()
    return [tok, tgt] + params, [loss] + ae.grad(loss, params), \
        [I((Tt, Bb), "int64", 1, 0, V), I((Tt, Bb), "int64", 2, 0, V)] + \
        [N(sh, seed=10 + k, scale=0.2) for k, sh in enumerate(shapes)]


# tensor/nnet front-end functions (tests/tensor/nnet/test_basic.py: test_softmax_with_bias,
# TestCrossEntropyCategorical1Hot :417, test_crossentropy_softmax_1hot_with_bias_dx): the
# logistic-regression tutorial graph — categorical_crossentropy(softmax(x W + b), y) with its
# gradients and an SGD update — and the activation helpers
for _dt, _tol in (("float64", 1e-12), ("float32", 2e-5)):
    def _mkLR(dt=_dt):
        import aesara.tensor.nnet as nn
        x, W, b, y = T(dt, (24, 10), "x"), T(dt, (10, 6), "W"), T(dt, (6,), "b"), at.lvector("y")
        p = nn.softmax(at.dot(x, W) + b)
        loss = nn.categorical_crossentropy(p, y).mean() + np.asarray(1e-3, dt) * (W ** 2).sum()
        gW, gb = ae.grad(loss, [W, b])
        lr = np.asarray(0.1, dt)
        return [x, W, b, y], [loss, W - lr * gW, b - lr * gb, at.argmax(p, axis=1),
                              nn.categorical_crossentropy(p, y), nn.relu(x), nn.relu(x, 0.1), nn.elu(x),
                              nn.softsign(x), nn.hard_sigmoid(x),
                              nn.binary_crossentropy(at.sigmoid(at.dot(x, W)), at.gt(at.dot(x, W), 0.3)).mean()], \
            [N((24, 10), dt, 1), N((10, 6), dt, 2, 0.3), N((6,), dt, 3, 0.1), I((24,), "int64", 4, 0, 6)]
    case(f"nnet_logreg_tutorial_{_dt}", rtol=_tol, atol=_tol)(_mkLR)


# Sort / ArgSort (tests/tensor/test_sort.py TestSort :50, test_argsort :186, test_argsort_grad):
# distinct keys (NumPy's default introsort leaves the order of ties unspecified), NaN last,
# every axis, negative axis, axis=None, gradients through the permutation
@case("sort_argsort", rtol=1e-12, atol=1e-12)
def _():
    x, t3, iv = at.dmatrix("x"), at.dtensor3("t3"), at.lvector("iv")
    xn = at.switch(at.gt(x, 1.5), np.nan, x)
    return [x, t3, iv], [at.sort(x), at.sort(x, axis=0), at.argsort(x, axis=1), at.argsort(x, axis=0),
                         at.sort(t3, axis=1), at.argsort(t3, axis=-1), at.sort(x, axis=None), at.argsort(x, axis=None),
                         at.sort(iv), at.argsort(iv), at.sort(xn, axis=1), at.sort(x.T[::2], axis=0),
                         ae.grad((at.sort(x, axis=1) * at.arange(11)).sum(), x),
                         x[at.arange(7)[:, None], at.argsort(x, axis=1)][:, :3]], \
        [N((7, 11), seed=1), N((3, 6, 5), seed=2), {"kind": "perm", "seed": 3, "shape": [9], "dtype": "int64", "n": 50}]


# Nonzero and boolean-mask indexing (tests/tensor/test_basic.py TestNonzero :4116,
# tests/tensor/test_subtensor.py test_boolean / test_adv_boolean :2450-2560): run-time sized results
@case("nonzero_and_masks", exact=True, ref_py=True)
def _():
    x, t3, v = at.lmatrix("x"), at.ltensor3("t3"), at.lvector("v")
    m1 = at.gt(v, 0)
    outs = list(at.nonzero(x)) + list(at.nonzero(t3 > 5)) + [at.flatnonzero(v), at.nonzero(x, return_matrix=True)]
    outs += [x[x > 0], x[at.lt(x, -50)] * 2, t3[t3 > 90], x[m1[:6]], x[m1[:6], 1:3], t3[:, at.gt(x[:5, :6], 0)],
             v[m1].sum(), at.set_subtensor(x[x > 0], 0), at.inc_subtensor(x[at.lt(x, 0)], 100),
             at.set_subtensor(t3[t3 > 0], v[0]), at.inc_subtensor(x[m1[:6]], x[0] * 0 + 7),
             x[(x > 200).nonzero()], x[at.eq(x, x)].sum()]
    return [x, t3, v], outs, [I((6, 8), "int64", seed=1, low=-99, high=99), I((4, 5, 6), "int64", seed=2, low=-99, high=99),
                              I((9,), "int64", seed=3, low=-5, high=5)]


@case("join_split_axes", exact=True)
def _():
    x, y, z = at.ltensor3("x"), at.ltensor3("y"), at.ltensor3("z")
    parts = at.split(at.join(2, x, y, z), [2, 5, 4], 3, axis=2)
    return [x, y, z], [at.join(0, x, x), at.join(-1, x, y, z), at.join(1, x.dimshuffle(0, 2, 1), z.dimshuffle(0, 2, 1)),
                       at.stack([x, x * 2], axis=1), parts[0] + 0, parts[1] * 1, parts[2] - 1], \
        [I((3, 4, 2), "int64", seed=1), I((3, 4, 5), "int64", seed=2), I((3, 4, 4), "int64", seed=3)]


@case("incsubtensor", exact=True)
def _():
    x, y, r = at.imatrix("x"), at.imatrix("y"), at.ivector("r")
    return [x, y, r], [at.set_subtensor(x[1:3], y), at.inc_subtensor(x[::4, 1], r[:3]),
                       at.inc_subtensor(x[2:4, :], y), at.set_subtensor(x[:, 0], 7)], \
        [I((9, 11), seed=1), I((2, 11), seed=2), I((5,), seed=3)]


@case("incsubtensor_f64", exact=True)
def _():
    x, y = at.dmatrix("x"), at.dvector("y")
    return [x, y], [at.inc_subtensor(x[3], y), at.set_subtensor(x[::2, ::3], 1.5)], \
        [N((8, 10), seed=1), N((10,), seed=2)]


@case("advsub1", exact=True)
def _():
    x, idx, idx32 = at.imatrix("x"), at.lvector("idx"), at.ivector("idx32")
    v = at.dvector("v")
    return [x, idx, idx32, v], [x[idx], x[idx32], v[idx], x.T[idx32]], \
        [I((13, 7), seed=1), I((20,), "int64", 2, -13, 13), I((5,), "int32", 3, -7, 7),
         N((13,), seed=4)]


# AdvancedIncSubtensor1's C thunk uses the NumPy-1 MapIter C-API (tensor/subtensor.py:2230) and does
# not compile against NumPy 2; the reference's Python perform (:2267) is the oracle here.
@case("advincsub1_int", exact=True, ref_py=True)
def _():
    x, y, idx = at.imatrix("x"), at.imatrix("y"), at.lvector("idx")
    return [x, y, idx], [at.inc_subtensor(x[idx], y), at.inc_subtensor(x[idx], 2)], \
        [I((13, 7), seed=1), I((20, 7), seed=2), I((20,), "int64", 3, -13, 13)]


@case("advsetsub1_dups", exact=True, ref_py=True)
def _():
    x, y, idx = at.imatrix("x"), at.imatrix("y"), at.lvector("idx")
    return [x, y, idx], [at.set_subtensor(x[idx], y)], \
        [I((13, 7), seed=1), I((20, 7), seed=2), I((20,), "int64", 3, -13, 13)]


@case("advincsub1_f32", rtol=1e-6, atol=1e-5, ref_py=True)
def _():
    x, y, idx = at.fmatrix("x"), at.fmatrix("y"), at.lvector("idx")
    return [x, y, idx], [at.inc_subtensor(x[idx], y)], \
        [N((50, 64), "float32", 1), N((300, 64), "float32", 2), I((300,), "int64", 3, 0, 50)]


# ---- round 5: what the reference's own test files found on the device (tests/reference_files.py) ----
@case("advinc_nodup_int", exact=True, ref_py=True)
def _():
    # AdvancedIncSubtensor(ignore_duplicates=True) (tensor/subtensor.py:2693): ``out[idx] += y`` —
    # read, add, sequential set: the LAST duplicate wins, its sum started from the original value
    from aesara.tensor.subtensor import advanced_inc_subtensor_nodup
    x, y = at.lmatrix("x"), at.lvector("y")
    i, j = at.lvector("i"), at.lvector("j")
    return [x, y, i, j], [advanced_inc_subtensor_nodup(x, y, i, j)], \
        [I((6, 5), "int64", 1), I((12,), "int64", 2), I((12,), "int64", 3, -6, 6), I((12,), "int64", 4, -5, 5)]


@case("gemm_bcast_z_f64", rtol=1e-12)
def _():
    # Gemm.perform (tensor/blas.py:995): z broadcast UP to the product's shape
    from aesara.tensor.blas import gemm_no_inplace
    z, x, y = at.dmatrix("z"), at.dmatrix("x"), at.dmatrix("y")
    a, b = at.dscalar("a"), at.dscalar("b")
    return [z, a, x, y, b], [gemm_no_inplace(z, a, x, y, b)], \
        [N((1, 40), seed=1), K(0.5, "float64"), N((33, 17), seed=2), N((17, 40), seed=3), K(0.25, "float64")]


@case("gemm_bcast_dot_f64", rtol=1e-12)
def _():
    # ... and ``z += a * dot(x, y)`` broadcasts the product up to z's shape
    from aesara.tensor.blas import gemm_no_inplace
    z, x, y = at.dmatrix("z"), at.dmatrix("x"), at.dmatrix("y")
    a, b = at.dscalar("a"), at.dscalar("b")
    return [z, a, x, y, b], [gemm_no_inplace(z, a, x, y, b)], \
        [N((35, 40), seed=1), K(0.5, "float64"), N((1, 17), seed=2), N((17, 40), seed=3), K(1.0, "float64")]


@case("join_mixed_dtypes", exact=True)
def _():
    # Join upcasts to the common dtype (tensor/basic.py:2214); the parts arrive in their own
    a, b, c = at.bmatrix("a"), at.imatrix("b"), at.matrix("c", dtype="int16")
    return [a, b, c], [at.join(1, a, b, c), at.join(0, a, c)], \
        [I((4, 3), "int8", 1), I((4, 5), "int32", 2, -1000, 1000), I((4, 3), "int16", 3, -300, 300)]


@case("arange_float32_fill", exact=True)
def _():
    # np.arange's fill rule in float32 (first, next, first + i * delta; product and sum each rounded)
    st, sp, se = at.fscalar("start"), at.fscalar("stop"), at.fscalar("step")
    return [st, sp, se], [at.arange(st, sp, se)], \
        [K(-5.0, "float32"), K(101.1, "float32"), K(1.2, "float32")]


@case("tile_8d_views", exact=True)
def _():
    # ``tile`` of a 4-d array: an 8-d DimShuffle view made contiguous for a Reshape, and the Sum over
    # such a view in its gradient — more non-mergeable dims than one kernel takes (6)
    x = at.TensorType("int64", shape=(None,) * 4)("x")
    t = at.tile(x, (2, 3, 2, 2))
    y = at.TensorType("int64", shape=(None,) * 8)("y")
    return [x, y], [t, y.dimshuffle(0, 2, 4, 6, 1, 3, 5, 7).sum(axis=(0, 1, 2, 3))], \
        [I((2, 3, 4, 5), "int64", 1), I((2, 3, 2, 4, 3, 2, 5, 2), "int64", 2)]


@case("scan_seq_products_two_row_counts", rtol=1e-12)
def _():
    # sequences of one Scan with DIFFERENT rows per step ([3, 4] next to [5, 6]): the hoisted
    # sequence products are stacked over time and unfolded per result (tests/scan/test_rewriting.py:741)
    s1, s2, s3 = at.dtensor3("s1"), at.dtensor3("s2"), at.dtensor3("s3")
    W, U_ = at.dmatrix("W"), at.dmatrix("U")
    init = at.dmatrix("init")

    def step(a, b, c, prev, W, U_):
        return prev + at.dot(at.dot(a, W) + c, at.dot(b, U_))
    h, _ = ae.scan(step, sequences=[s1, s2, s3], outputs_info=init, non_sequences=[W, U_])
    return [s1, s2, s3, W, U_, init], [h[-1], h], \
        [N((6, 3, 4), seed=1), N((6, 5, 6), seed=2), N((6, 3, 5), seed=3), N((4, 5), seed=4),
         N((6, 7), seed=5), N((3, 7), seed=6)]


@case("alloc_join", exact=True)
def _():
    v, m = at.dvector("v"), at.dmatrix("m")
    n = at.lscalar("n")
    return [v, m, n], [at.alloc(v, n, v.shape[0]), at.zeros((n, 3)) + 1.0,
                       at.concatenate([m, m * 2], axis=0), at.concatenate([m, m[:, :2]], axis=1),
                       at.alloc(np.float64(2.5), 2, n), m.reshape((m.shape[1], m.shape[0])),
                       m.T.reshape((-1,)), m.flatten()], \
        [N((5,), seed=1), N((4, 5), seed=2), K(3, "int64")]


# ---------------------------------------------------------------------------------------
# Scan (tests/scan/test_basic.py style restatements)
# ---------------------------------------------------------------------------------------
@case("scan_cumsum", rtol=1e-12, atol=1e-12)
def _():
    x, s0 = at.dmatrix("x"), at.dvector("s0")
    res, _ = ae.scan(lambda x_t, s: s + x_t, sequences=[x], outputs_info=[s0])
    return [x, s0], [res, res[-1]], [N((20, 7), seed=1), N((7,), seed=2)]


@case("scan_taps", rtol=1e-12, atol=1e-12)
def _():
    x, init, w = at.dvector("x"), at.dvector("init"), at.dscalar("w")
    res, _ = ae.scan(lambda x_t, a, b, w: w * a + 0.5 * b + x_t, sequences=[x],
                     outputs_info=[dict(initial=init, taps=[-2, -1])], non_sequences=[w])
    return [x, init, w], [res], [N((15,), seed=1, scale=0.1), N((2,), seed=2), K(0.7, "float64")]


# programmatically varied Scans (tests/scan/test_basic.py: test_using_taps_sequence :1012,
# test_past_future_taps_shared :1166, test_backwards :713, test_multiple_outs_taps :1330, n_steps,
# gradients through all of them): sequence taps, output taps, nit-sot outputs, non-sequences,
# go_backwards, explicit n_steps, matrix states — and d(sum of everything)/d(inputs)
def _scan_variant(k):
    def mk():
        rng = np.random.default_rng(900 + k)
        H = 4
        x, z, W, h0 = at.dmatrix("x"), at.dmatrix("z"), at.dmatrix("W"), at.dmatrix("h0")
        seq_taps = [[0], [0, -1], [-1, 1], [0, 2]][k % 4]
        out_taps = [[-1], [-1, -2], [-1, -3]][k % 3]
        two_seq = k % 2 == 1
        backwards = k % 5 == 2
        with_nit = k % 3 != 1
        fixed_steps = k % 4 == 3

        def step(*a):
            a = list(a)
            xs = [a.pop(0) for _ in seq_taps]
            zs = [a.pop(0)] if two_seq else []
            hs = [a.pop(0) for _ in out_taps]
            Wm = a.pop(0)
            pre = sum(xs[1:], xs[0]) * 0.5 + (zs[0] * 0.25 if zs else 0)
            h = at.tanh(at.dot(hs[0], Wm) + pre) + sum((0.1 * (q + 1) * t for q, t in enumerate(hs[1:])), 0)
            return [h, (h * h).sum() + pre.max()] if with_nit else h
        seqs = [dict(input=x, taps=seq_taps)] + ([z] if two_seq else [])
        outs_info = [dict(initial=h0[:len(set(out_taps)) if out_taps == [-1] else -min(out_taps)],
                          taps=out_taps)] + ([None] if with_nit else [])
        if out_taps == [-1]:
            outs_info[0] = h0[0]
        kw = dict(n_steps=5) if fixed_steps else {}
        res, _ = ae.scan(step, sequences=seqs, outputs_info=outs_info, non_sequences=[W],
                         go_backwards=backwards, **kw)
        res = res if isinstance(res, list) else [res]
        cost = sum((r ** 2).sum() for r in res)
        grads = ae.grad(cost, [x, W, h0])
        return [x, z, W, h0], res + grads, [N((9, H), seed=1, scale=0.5), N((9, H), seed=2, scale=0.5),
                                            N((H, H), seed=3, scale=0.4), N((3, H), seed=4, scale=0.3)]
    return mk


for _k in range(12):
    case(f"scan_variant_{_k}", rtol=1e-10, atol=1e-11)(_scan_variant(_k))


@case("scan_nested_with_grad", rtol=1e-11, atol=1e-12)
def _():
    # a Scan whose step runs another Scan (tests/scan/test_basic.py test_inner_scan :2306 /
    # nested gradients), truncate_gradient, strict non-sequences, integer counters
    x, W, v = at.dmatrix("x"), at.dmatrix("W"), at.dvector("v")

    def outer(row, acc):
        inner, _ = ae.scan(lambda e, st: st * 0.5 + e, sequences=[row], outputs_info=[at.zeros_like(row[0])])
        return acc + inner[-1] * row
    r, _ = ae.scan(outer, sequences=[x], outputs_info=[at.zeros_like(x[0])])
    h, _ = ae.scan(lambda a, hh, Wm: at.tanh(at.dot(hh, Wm) + a), sequences=[x], outputs_info=[at.zeros_like(x[0])],
                   non_sequences=[W], truncate_gradient=3)
    hs, _ = ae.scan(lambda a, hh, Wm, vv: at.tanh(at.dot(hh, Wm) + a * vv), sequences=[x], outputs_info=[v],
                    non_sequences=[W, v], strict=True)
    (cnt, val), _ = ae.scan(lambda i, c, q: (c + 1, q * 2 + i), sequences=[at.arange(5)],
                            outputs_info=[at.as_tensor(np.int64(0)), at.as_tensor(np.int64(1))])
    return [x, W, v], [r[-1], ae.grad(r[-1].sum(), x)] + ae.grad(h[-1].sum(), [x, W]) + \
        [hs, ae.grad(hs.sum(), v), cnt, val], [N((6, 4), seed=1), N((4, 4), seed=2, scale=0.3), N((4,), seed=3)]


@case("scan_nitsot_map", rtol=1e-12, atol=1e-12)
def _():
    x, W = at.dmatrix("x"), at.dmatrix("W")
    res, _ = ae.scan(lambda x_t, W: at.tanh(at.dot(W, x_t)), sequences=[x], non_sequences=[W])
    return [x, W], [res], [N((12, 6), seed=1), N((6, 6), seed=2)]


@case("gemv_runtime_beta_zero_ignores_y", rtol=1e-13, atol=0)
def _():
    """``a * y + dot(A, x)`` with a RUN-TIME scalar a (-> Gemv(y, 1, A, x, a)): a == 0 never reads y
    (tensor/blas.py:236 Gemv.perform, BLAS xGEMV) — the NaNs of y do not reach the result
    (tests/tensor/test_blas_c.py:146); the second output keeps them (a != 0)."""
    A, x, x2, y, a, b = at.dmatrix("A"), at.dvector("x"), at.dvector("x2"), at.dvector("y"), at.dscalar("a"), at.dscalar("b")
    return [A, x, x2, y, a, b], [a * y + at.dot(A, x), b * y + at.dot(A, x2)], \
        [N((33, 17), seed=1), N((17,), seed=2), N((17,), seed=4),
         {"kind": "normal_with_nan", "seed": 3, "shape": [33], "dtype": "float64"}, K(0.0, "float64"), K(0.5, "float64")]


@case("scan_embedding_lookup_in_step", rtol=1e-12, atol=1e-12)
def _():
    """An RNN whose step looks its input up itself — ``E[idx_t]`` by the step's index (a vector
    state) — plus a step that builds ``one_hot(idx_t)`` with set_subtensor: index glue on sequence
    rows inside a recurrent step (fusion.push_out_sequence_glue restates it over whole sequences
    in front of the Scan; the loop that is left is the plain ``tanh(x_t + h U)`` recurrence)."""
    idx, E, U, h0 = at.lvector("idx"), at.dmatrix("E"), at.dmatrix("U"), at.dvector("h0")

    def step(i, h, E, U):
        return at.tanh(E[i] + at.dot(h, U))
    hs, _ = ae.scan(step, sequences=[idx], outputs_info=[h0], non_sequences=[E, U])

    def step2(i, k, h, E, U):
        e = at.set_subtensor(at.zeros_like(E[0])[k], 1.0)
        return at.tanh(E[i + 1] * 0.5 + e + at.dot(h, U))
    hs2, _ = ae.scan(step2, sequences=[idx, idx % 6], outputs_info=[h0], non_sequences=[E, U])
    return [idx, E, U, h0], [hs, hs2[-1], ae.grad(hs.sum(), E)], \
        [I((11,), "int64", seed=1, low=-9, high=9), N((10, 6), seed=2), N((6, 6), seed=3, scale=0.4), N((6,), seed=4)]


@case("scan_grad_of_row_lookup", rtol=1e-12, atol=1e-12)
def _():
    """tests/scan/test_basic.py:3138 (test_grad_bug_disconnected_input): the gradient of a map that
    looks rows of W up by a sequence of indices — the gradient Scan's ONLY output is the scatter
    accumulator ``dW[i_t] += g_t`` over sequence rows: with the scatter taken out no loop is left."""
    v, W = at.lvector("v"), at.dmatrix("W")
    y, _ = ae.scan(lambda i, W: W[i] * 2.0, sequences=v, outputs_info=None, non_sequences=W)
    return [v, W], [ae.grad((y ** 2).sum(), W), y], [I((7,), "int64", seed=1, low=-4, high=4), N((4, 3), seed=2)]


@case("scan_embedding_lookup_batch_f32", rtol=2e-5, atol=2e-6, ref_py=True)
def _():
    """The same for a BATCH of recurrences: ``E[idx_t]`` with the step's index VECTOR
    (AdvancedSubtensor1 per step -> one gather over T * B indices), float32 matrix state."""
    idx, E, U, h0 = at.lmatrix("idx"), at.fmatrix("E"), at.fmatrix("U"), at.fmatrix("h0")

    def step(i, h, E, U):
        return at.tanh(E[i] + at.dot(h, U))
    hs, _ = ae.scan(step, sequences=[idx], outputs_info=[h0], non_sequences=[E, U])
    cost = (hs[-1] ** 2).sum()
    return [idx, E, U, h0], [hs, ae.grad(cost, U), ae.grad(cost, E)], \
        [I((9, 16), "int64", seed=1, low=0, high=40), N((40, 64), "float32", seed=2),
         N((64, 64), "float32", seed=3, scale=0.1), N((16, 64), "float32", seed=4)]


@case("scan_map_jacobian_rows", rtol=1e-12, atol=1e-12)
def _():
    """``gradient.jacobian`` (gradient.py:1930): a Scan over ``arange(n)`` with no recurrence whose
    step takes row i of the gradient — the all-rows restatement (fusion.batch_map_step)."""
    x, W = at.dvector("x"), at.dmatrix("W")
    y = at.tanh(at.dot(W, x)) * x.sum()
    return [x, W], [ae.gradient.jacobian(y, x)], [N((5,), seed=1), N((5, 5), seed=2, scale=0.5)]


@case("scan_map_hessian_unit_vectors", rtol=1e-12, atol=1e-12)
def _():
    """``gradient.hessian`` (gradient.py:2027) of a scalar cost: the step differentiates one entry of
    the gradient; plus a map whose step builds the unit vector e_i with ``set_subtensor`` and an
    ``inc_subtensor`` on a per-step value."""
    x, A = at.dvector("x"), at.dmatrix("A")
    cost = at.sum(at.tanh(at.dot(A, x)) ** 2) + at.sum(x ** 3)
    H = ae.gradient.hessian(cost, x)

    def step(i, x, A):
        e = at.set_subtensor(at.zeros_like(x)[i], 1.0)
        u = at.dot(A, e) * x
        return at.inc_subtensor(u[1], x[i]), at.sum(u * e)
    (cols, diag), _ = ae.scan(step, sequences=[at.arange(x.shape[0])], non_sequences=[x, A])
    return [x, A], [H, cols, diag], [N((4,), seed=3), N((4, 4), seed=4, scale=0.6)]


@case("scan_map_rows_reduce_broadcast", rtol=1e-12, atol=1e-12)
def _():
    """Maps without recurrence over matrix rows: reductions, DimShuffle, an invariant operand, a
    0-d result, a constant-per-step output, a negative index sequence into an invariant table, the
    last rows only (``[-3:]``: scan_save_mem keeps three)."""
    M, b, idx = at.dmatrix("M"), at.dvector("b"), at.lvector("idx")

    def step(r, k, b, M):
        z = at.exp(r - r.max()) * b
        return z / z.sum(), (r.dimshuffle(0, "x") * b.dimshuffle("x", 0)).sum(axis=0), M[k, 1:] * 2.0, r.sum()
    (sm, outer, picked, tot), _ = ae.scan(step, sequences=[M, idx], non_sequences=[b, M])
    return [M, b, idx], [sm, outer[-3:], picked, tot], [N((9, 6), seed=5), N((6,), seed=6),
                                                     I((9,), "int64", seed=7, low=-9, high=9)]


@case("scan_nitsot_value_and_its_view", rtol=1e-12, atol=1e-12)
def _():
    """Two nit-sot outputs that are the same value in two shapes (a vector and the [1, n] row view of
    it, DimShuffle 'x', 0) next to the recurrent output — what a gradient Scan hands out when a
    per-step vector is needed both ways; exercises the launch list's write-into-the-output-row
    targets when one output is a view of another."""
    x, W = at.dmatrix("x"), at.dmatrix("W")

    def step(x_t, h, W):
        u = at.tanh(at.dot(h, W) + x_t)
        return u * 0.5 + h * 0.5, u, u.dimshuffle("x", 0)
    (hs, us, urows), _ = ae.scan(step, sequences=[x], outputs_info=[at.zeros_like(x[0]), None, None],
                                 non_sequences=[W])
    return [x, W], [hs, us, urows, (urows * 2.0).sum(axis=1)], [N((7, 5), seed=1), N((5, 5), seed=2, scale=0.4)]


@case("scan_nitsot_view_then_value", rtol=1e-12, atol=1e-12)
def _():
    """The same with the row view handed out BEFORE the value it is a view of, and the value twice."""
    x, W = at.dmatrix("x"), at.dmatrix("W")

    def step(x_t, h, W):
        u = at.tanh(at.dot(h, W) + x_t)
        return u * 0.5 + h * 0.5, u.dimshuffle("x", 0), u, u
    (hs, urows, us, us2), _ = ae.scan(step, sequences=[x], outputs_info=[at.zeros_like(x[0]), None, None, None],
                                      non_sequences=[W])
    return [x, W], [hs, urows, us, us2 * 3.0], [N((7, 5), seed=1), N((5, 5), seed=2, scale=0.4)]


@case("scan_two_outputs", rtol=1e-12, atol=1e-12)
def _():
    x, a0, b0 = at.dvector("x"), at.dscalar("a0"), at.dscalar("b0")
    res, _ = ae.scan(lambda x_t, a, b: [a * 0.9 + x_t, b + a], sequences=[x],
                     outputs_info=[a0, b0])
    return [x, a0, b0], [res[0], res[1][-1]], [N((25,), seed=1), K(0.5, "float64"),
                                              K(-1.0, "float64")]


# do-while scans (tests/scan/test_basic.py test_while0 / test_while1 / test_while_infershape)
@case("scan_while_cumsum", rtol=1e-12, atol=1e-12)
def _():
    from aesara.scan.utils import until
    x = at.dvector("x")
    res, _ = ae.scan(lambda x_t, s: (s + x_t, until(s + x_t > 3.0)), sequences=[x],
                     outputs_info=[at.as_tensor_variable(np.float64(0.0))])
    return [x], [res, res[-1], res.shape[0]], [U((40,), seed=3, low=0.1, high=0.9)]


@case("scan_while_never_stops", rtol=1e-12, atol=1e-12)
def _():
    from aesara.scan.utils import until
    x = at.dvector("x")
    res, _ = ae.scan(lambda x_t, s: (s + x_t, until(s + x_t > 1e9)), sequences=[x],
                     outputs_info=[at.as_tensor_variable(np.float64(0.0))])
    return [x], [res], [U((17,), seed=4)]


@case("scan_while_nitsot_matrix", rtol=1e-12, atol=1e-12)
def _():
    from aesara.scan.utils import until
    x, W, h0 = at.dmatrix("x"), at.dmatrix("W"), at.dvector("h0")

    def step(x_t, h, W):
        hn = at.tanh(at.dot(W, h) + x_t)
        return [hn, (hn ** 2).sum()], until((hn ** 2).sum() > 2.5)
    (hs, en), _ = ae.scan(step, sequences=[x], outputs_info=[h0, None], non_sequences=[W])
    return [x, W, h0], [hs, en, hs.shape[0]], \
        [N((30, 6), seed=1), N((6, 6), seed=2, scale=0.5), K(0.0, "float64", (6,))]


# full reductions inside the step (scan_perform.pyx:309-541 runs them in the same loop; the do-while
# condition of a VECTOR state is a reduction, :424-426): tests/scan/test_basic.py TestGradUntil
# test_grad_until_ndim_greater_one :2391 (``until(at_all(x > u))``), power-iteration-like
# normalisation, a convergence loop, a per-step energy next to a vector state, a 0-d running total
@case("scan_red_normalise", rtol=1e-12, atol=1e-12)
def _():
    x0, a = at.dvector("x0"), at.dvector("a")
    res, _ = ae.scan(lambda x, a: x * a / at.abs(x * a).sum(), outputs_info=[x0], non_sequences=[a],
                     n_steps=7)
    return [x0, a], [res, res[-1]], [U((40,), seed=1, low=0.5, high=1.5), U((40,), seed=2, low=0.5, high=1.5)]


@case("scan_red_normalise_wide_f32", rtol=2e-5, atol=1e-6)
def _():
    # 300 elements: five wavefronts, the partials cross through LDS
    x0, a = at.fvector("x0"), at.fvector("a")
    res, _ = ae.scan(lambda x, a: x * a / at.sqrt((x * a * x * a).sum()), outputs_info=[x0],
                     non_sequences=[a], n_steps=9)
    return [x0, a], [res], [U((300,), "float32", seed=1, low=0.5, high=1.5),
                            U((300,), "float32", seed=2, low=0.5, high=1.5)]


@case("scan_red_until_all", rtol=1e-12, atol=1e-12)
def _():
    from aesara.scan.utils import until
    X, thr = at.dmatrix("X"), at.dscalar("thr")
    r, _ = ae.scan(lambda x, u: (x * x, until(at.all(x > u))), sequences=X, non_sequences=[thr])
    return [X, thr], [r, r.shape[0]], [
        {"kind": "const_list", "shape": [15, 5], "dtype": "float64",
         "values": [float(i) for i in range(15) for _ in range(5)]}, K(5.0, "float64")]


@case("scan_red_newton_until", rtol=1e-12, atol=1e-12)
def _():
    from aesara.scan.utils import until
    x0, a = at.dvector("x0"), at.dvector("a")

    def step(x, a):
        xn = 0.5 * (x + a / x)
        return xn, until(at.max(at.abs(xn - x)) < 1e-9)
    r, _ = ae.scan(step, outputs_info=[x0], non_sequences=[a], n_steps=50)
    return [x0, a], [r, r[-1], r.shape[0]], [U((33,), seed=3, low=1.0, high=2.0), U((33,), seed=4, low=1.0, high=9.0)]


@case("scan_red_energy", rtol=1e-12, atol=1e-12)
def _():
    s, h0 = at.dmatrix("s"), at.dvector("h0")
    (hs, en), _ = ae.scan(lambda s_t, h: (0.9 * h + s_t, ((0.9 * h + s_t) ** 2).sum()), sequences=[s],
                          outputs_info=[h0, None])
    return [s, h0], [hs, en], [N((20, 70), seed=1), N((70,), seed=2)]


@case("scan_red_running_total", rtol=1e-12, atol=1e-12)
def _():
    # a 0-d recurrent state fed by a reduction of the step's vector, next to a vector state; only
    # the last total is used (scan_save_mem: a circular buffer of the 0-d state)
    s, h0 = at.dmatrix("s"), at.dvector("h0")
    (hs, tot), _ = ae.scan(lambda s_t, h, tot: (at.tanh(h + s_t), tot + (h * s_t).sum()), sequences=[s],
                           outputs_info=[h0, at.as_tensor_variable(np.float64(0.0))])
    return [s, h0], [hs[-1], tot[-1], tot], [N((25, 130), seed=5), N((130,), seed=6)]


@case("scan_red_rnn_normalised", rtol=1e-11, atol=1e-12)
def _():
    # power-iteration-like recurrence: the new state is divided by its own norm (a reduction whose
    # result feeds the SAME step's state), the norm is handed out per step
    x, W, h0 = at.dmatrix("x"), at.dmatrix("W"), at.dvector("h0")

    def step(x_t, h, W):
        z = at.tanh(at.dot(W, h) + x_t)
        nrm = at.sqrt((z ** 2).sum())
        return [z / nrm, nrm]
    (hs, nr), _ = ae.scan(step, sequences=[x], outputs_info=[h0, None], non_sequences=[W])
    return [x, W, h0], [hs, nr], [N((24, 70), seed=1), N((70, 70), seed=2, scale=0.3), N((70,), seed=3)]


@case("scan_red_rnn_until_f32", rtol=2e-5, atol=1e-6)
def _():
    # do-while over a vector state with a dot in the step: stop when the state's largest element
    # passes a threshold (scan_perform.pyx:424-426); trip count, truncated outputs
    from aesara.scan.utils import until
    x, W, h0 = at.fmatrix("x"), at.fmatrix("W"), at.fvector("h0")

    def step(x_t, h, W):
        hn = 0.9 * h + at.tanh(at.dot(W, h) + x_t) * 0.5
        return [hn, hn.max()], until(hn.max() > 1.25)
    (hs, mx), _ = ae.scan(step, sequences=[x], outputs_info=[h0, None], non_sequences=[W])
    return [x, W, h0], [hs, mx, hs.shape[0]], [
        N((40, 96), "float32", seed=4, scale=0.5), N((96, 96), "float32", seed=5, scale=0.1),
        K(0.0, "float32", (96,))]


@case("scan_red_int_minmax", exact=True)
def _():
    # integer reductions (exact): per-step minimum and maximum of a running integer vector, and a
    # bool ``any`` deciding a scale factor
    x, v0 = at.lmatrix("x"), at.lvector("v0")

    def step(x_t, v):
        vn = v + x_t
        big = at.any(vn > 40)
        return [at.switch(big, vn // 2, vn), vn.min(), vn.max()]
    (vs, lo, hi), _ = ae.scan(step, sequences=[x], outputs_info=[v0, None, None])
    return [x, v0], [vs, lo, hi], [I((18, 77), "int64", seed=1, low=-5, high=12), I((77,), "int64", seed=2)]


# gradients through Scan: Scan.L_op (scan/op.py:2379) builds a reversed Scan with mit-mot
# accumulators (tests/scan/test_basic.py test_grad_one_output / test_grad_multiple_outs_taps)
@case("scan_grad_rnn", rtol=1e-11, atol=1e-11)
def _():
    x, h0, W = at.dmatrix("x"), at.dvector("h0"), at.dmatrix("W")
    hs, _ = ae.scan(lambda x_t, h, W: at.tanh(at.dot(h, W) + x_t), sequences=[x],
                    outputs_info=[h0], non_sequences=[W])
    cost = (hs ** 2).sum()
    return [x, h0, W], [cost] + list(ae.grad(cost, [W, h0, x])), \
        [N((9, 5), seed=1), N((5,), seed=2), N((5, 5), seed=3, scale=0.4)]


@case("scan_grad_taps", rtol=1e-11, atol=1e-11)
def _():
    x, init, w = at.dvector("x"), at.dvector("init"), at.dscalar("w")
    res, _ = ae.scan(lambda x_t, a, b, w: at.tanh(w * a + 0.5 * b) + x_t, sequences=[x],
                     outputs_info=[dict(initial=init, taps=[-2, -1])], non_sequences=[w])
    cost = (res * res).sum()
    return [x, init, w], [cost] + list(ae.grad(cost, [x, init, w])), \
        [N((11,), seed=1, scale=0.3), N((2,), seed=2), K(0.7, "float64")]


@case("scan_grad_taps_wide", rtol=1e-10, atol=1e-10)
def _():
    """The gradient of a two-tap element-wise recurrence over vectors: Scan.L_op propagates it
    through a mit-mot output with taps [0, 2, 1] -> [2, 1] (scan/op.py:2379) — 130 independent
    elements, 37 steps (more than the kernel's read-ahead of 8)."""
    x, init, w = at.dmatrix("x"), at.dmatrix("init"), at.dvector("w")
    res, _ = ae.scan(lambda x_t, a, b, w: at.tanh(w * a + 0.5 * b) + x_t, sequences=[x],
                     outputs_info=[dict(initial=init, taps=[-2, -1])], non_sequences=[w])
    cost = (res * res).sum()
    return [x, init, w], [cost] + list(ae.grad(cost, [x, init, w])), \
        [N((37, 130), seed=1, scale=0.3), N((2, 130), seed=2), N((130,), seed=3, scale=0.5)]


@case("scan_grad_taps13", rtol=1e-10, atol=1e-10)
def _():
    """Taps [-3, -1] and a second output with one tap: two mit-mot groups with different windows
    in the gradient Scan."""
    x, a0, b0 = at.dmatrix("x"), at.dmatrix("a0"), at.dvector("b0")

    def step(x_t, a3, a1, b):
        an = at.tanh(0.6 * a1 - 0.3 * a3 * b) + x_t
        bn = at.sigmoid(b + 0.2 * an)
        return an, bn
    (as_, bs), _ = ae.scan(step, sequences=[x], outputs_info=[dict(initial=a0, taps=[-3, -1]), b0])
    cost = (as_ * as_).sum() + (bs[-1] * bs[-1]).sum()
    return [x, a0, b0], [cost] + list(ae.grad(cost, [x, a0, b0])), \
        [N((19, 5), seed=1, scale=0.3), N((3, 5), seed=2), N((5,), seed=3, scale=0.5)]


@case("scan_grad_last_state_f32", rtol=2e-5, atol=2e-5)
def _():
    x, h0, W, U_ = at.fmatrix("x"), at.fvector("h0"), at.fmatrix("W"), at.fmatrix("U")
    hs, _ = ae.scan(lambda x_t, h, W, U_: at.tanh(at.dot(x_t, W) + at.dot(h, U_)),
                    sequences=[x], outputs_info=[h0], non_sequences=[W, U_])
    cost = hs[-1].sum()
    return [x, h0, W, U_], [cost] + list(ae.grad(cost, [W, U_, h0])), \
        [N((12, 8), "float32", 1), N((16,), "float32", 2), N((8, 16), "float32", 3, 0.3),
         N((16, 16), "float32", 4, 0.25)]


# CheckAndRaise / SpecifyShape views (raise_op.py:28, tensor/shape.py:376)
@case("assert_specify_shape", rtol=1e-13, atol=1e-13)
def _():
    from aesara.raise_op import Assert
    x, n = at.dmatrix("x"), at.lscalar("n")
    a = Assert("n must be positive")(x, at.gt(n, 0))
    b = at.specify_shape(x, (None, 5))
    return [x, n], [a + 1.0, b * 2.0, at.specify_shape(x, (n, None)).sum(axis=0)], \
        [N((4, 5), seed=1), K(4, "int64")]



def _gru(dt, T_, H, B_, tol):
    def mk():
        x = T(dt, (2, 2) if B_ == 1 else (2, 2, 2), "x")
        h0 = T(dt, (2,) if B_ == 1 else (2, 2), "h0")
        Ws = [T(dt, (2, 2), n) for n in ("Wz", "Uz", "Wr", "Ur", "Wh", "Uh")]

        def step(x_t, h, Wz, Uz, Wr, Ur, Wh, Uh):
            z = at.sigmoid(at.dot(x_t, Wz) + at.dot(h, Uz))
            r = at.sigmoid(at.dot(x_t, Wr) + at.dot(h, Ur))
            hh = at.tanh(at.dot(x_t, Wh) + at.dot(r * h, Uh))
            return (1 - z) * h + z * hh
        hs, _ = ae.scan(step, sequences=[x], outputs_info=[h0], non_sequences=Ws)
        xs = (T_, H) if B_ == 1 else (T_, B_, H)
        hsz = (H,) if B_ == 1 else (B_, H)
        return [x, h0] + Ws, [hs, hs[-1]], \
            [N(xs, dt, 4, 0.1), K(0.0, dt, hsz)] + \
            [N((H, H), dt, 5 + k, 1.0 / np.sqrt(H)) for k in range(6)]
    return mk


def _gru_bptt(dt, T_, H, B_):
    """BASELINE config 4's recurrence under aesara.grad: loss = sum(h_T ** 2) + mean(hs), gradients
    wrt the six weight matrices and h0 — the forward Scan keeps every state, the gradient Scan
    (mit-mot accumulators, reversed sequences) runs behind it (scan/op.py:2379 Scan.L_op)."""
    def mk():
        x = T(dt, (2, 2) if B_ == 1 else (2, 2, 2), "x")
        h0 = T(dt, (2,) if B_ == 1 else (2, 2), "h0")
        Ws = [T(dt, (2, 2), n) for n in ("Wz", "Uz", "Wr", "Ur", "Wh", "Uh")]

        def step(x_t, h, Wz, Uz, Wr, Ur, Wh, Uh):
            z = at.sigmoid(at.dot(x_t, Wz) + at.dot(h, Uz))
            r = at.sigmoid(at.dot(x_t, Wr) + at.dot(h, Ur))
            hh = at.tanh(at.dot(x_t, Wh) + at.dot(r * h, Uh))
            return (1 - z) * h + z * hh
        hs, _ = ae.scan(step, sequences=[x], outputs_info=[h0], non_sequences=Ws)
        loss = (hs[-1] ** 2).sum() + hs.mean()
        grads = ae.grad(loss, Ws + [h0])
        xs = (T_, H) if B_ == 1 else (T_, B_, H)
        hsz = (H,) if B_ == 1 else (B_, H)
        return [x, h0] + Ws, [loss] + grads, \
            [N(xs, dt, 4, 0.3), N(hsz, dt, 3, 0.5)] + \
            [N((H, H), dt, 5 + k, 1.0 / np.sqrt(H)) for k in range(6)]
    return mk


@case("lstm_bptt_vec_f32", rtol=3e-4, atol=3e-5)
def _():
    """LSTM with a vector state under aesara.grad (two recurrent states -> two mit-mot groups in
    the gradient Scan), separate weight matrices per gate, loss on the last hidden state."""
    x, h0, c0 = at.fmatrix("x"), at.fvector("h0"), at.fvector("c0")
    names = ("Wi", "Ui", "Wf", "Uf", "Wo", "Uo", "Wg", "Ug")
    Ws = [at.fmatrix(n) for n in names]
    bf = at.fvector("bf")

    def step(x_t, h, c, Wi, Ui, Wf, Uf, Wo, Uo, Wg, Ug, bf):
        i = at.sigmoid(at.dot(x_t, Wi) + at.dot(h, Ui))
        f = at.sigmoid(at.dot(x_t, Wf) + at.dot(h, Uf) + bf)
        o = at.sigmoid(at.dot(x_t, Wo) + at.dot(h, Uo))
        g = at.tanh(at.dot(x_t, Wg) + at.dot(h, Ug))
        c2 = f * c + i * g
        return o * at.tanh(c2), c2
    (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[h0, c0], non_sequences=Ws + [bf])
    loss = (hs[-1] ** 2).sum() + cs.mean()
    grads = ae.grad(loss, Ws + [bf, h0])
    D, H = 12, 16
    return [x, h0, c0] + Ws + [bf], [loss] + grads, \
        [N((9, D), "float32", 4, 0.5), N((H,), "float32", 3, 0.5), N((H,), "float32", 2, 0.5)] + \
        [N((D if k % 2 == 0 else H, H), "float32", 5 + k, 1.0 / np.sqrt(D if k % 2 == 0 else H))
         for k in range(8)] + [N((H,), "float32", 20, 0.2)]


case("gru_bptt_b1_f32", rtol=2e-4, atol=2e-5)(_gru_bptt("float32", 12, 20, 1))
case("gru_bptt_b4_f64", rtol=1e-9, atol=1e-10)(_gru_bptt("float64", 9, 12, 4))
case("gru_bptt_b4_f32", rtol=3e-4, atol=3e-5)(_gru_bptt("float32", 10, 16, 4))
case("cfg4_gru_b1_f32", rtol=1e-5, atol=1e-5)(_gru("float32", 32, 64, 1, 1e-5))
case("cfg4_gru_b8_f32", rtol=1e-5, atol=1e-5)(_gru("float32", 16, 64, 8, 1e-5))
case("gru_b1_f64", rtol=1e-11, atol=1e-11)(_gru("float64", 10, 24, 1, 1e-11))
# batch a multiple of 16, state a multiple of 64: the shapes at which the persistent matrix kernel
# computes the sequence products x_t @ W itself (scan_persist_mat.xfold_windows) — three products
# over the two windows of a GRU step, one of them a step ahead
case("xfold_gru_b16_f32", rtol=1e-5, atol=1e-5)(_gru("float32", 12, 64, 16, 1e-5))


@case("xfold_rnn_b16_f32", rtol=1e-5, atol=1e-5)
def _():
    """Elman recurrence with a batch: ONE fetching phase per step, so its sequence product runs a
    whole step ahead (x_{t+1} in the window of step t, x_{t+2} requested behind it)."""
    x, h0, W, U = at.ftensor3("x"), at.fmatrix("h0"), at.fmatrix("W"), at.fmatrix("U")

    def step(x_t, h, W, U):
        return at.tanh(at.dot(x_t, W) + at.dot(h, U))
    hs, _ = ae.scan(step, sequences=[x], outputs_info=[h0], non_sequences=[W, U])
    return [x, h0, W, U], [hs, hs[-1]], \
        [N((9, 16, 64), "float32", 4, 0.5), N((16, 64), "float32", 3, 0.5),
         N((64, 64), "float32", 5, 0.125), N((64, 64), "float32", 6, 0.125)]



@case("sm_taps_b16_f32", rtol=1e-5, atol=1e-5)
def _():
    """A batch of recurrences with taps [-1, -2] (scan_perform.pyx:321-340 hands the step one row
    per tap): h_t = tanh(h_{t-1} @ U + 0.5 * h_{t-2} + x_t) — the older tap is element-wise, so it
    stays in the registers of the tile element's owner (matrix-state persistent kernel)."""
    x, h0, U = at.ftensor3("x"), at.ftensor3("h0"), at.fmatrix("U")

    def step(x_t, h2, h1, U):
        return at.tanh(at.dot(h1, U) + np.float32(0.5) * h2 + x_t)
    hs, _ = ae.scan(step, sequences=[x], outputs_info=[dict(initial=h0, taps=[-2, -1])], non_sequences=[U])
    return [x, h0, U], [hs, hs[-1]], \
        [N((9, 16, 64), "float32", 4, 0.5), N((2, 16, 64), "float32", 3, 0.5), N((64, 64), "float32", 5, 0.125)]


@case("sp_taps_bptt_f64", rtol=1e-9, atol=1e-10)
def _():
    """Two-tap recurrence on a vector state and its gradient: h_t = tanh(h_{t-1} @ U + 0.5 h_{t-2} + x_t).
    Scan.L_op propagates through a mit-mot output with taps [0, 2, 1] -> [2, 1] (scan/op.py:2379)
    AND accumulates the weight gradient inside the gradient Scan (a sit-sot output; PushOutDot1 does
    not lift it for this tap set): the forward Scan runs persistent, the gradient Scan on the launch
    list."""
    x, h0, U = at.dmatrix("x"), at.dmatrix("h0"), at.dmatrix("U")

    def step(x_t, h2, h1, U):
        return at.tanh(at.dot(h1, U) + 0.5 * h2 + x_t)
    hs, _ = ae.scan(step, sequences=[x], outputs_info=[dict(initial=h0, taps=[-2, -1])], non_sequences=[U])
    cost = (hs * hs).sum()
    return [x, h0, U], [cost] + list(ae.grad(cost, [x, h0, U])), \
        [N((11, 64), seed=4, scale=0.5), N((2, 64), seed=3, scale=0.5), N((64, 64), seed=5, scale=0.125)]


@case("sm_taps_bptt_b16_f32", rtol=2e-4, atol=2e-5)
def _():
    """The same with a batch (matrix state): forward = taps in the owner's registers (persistent),
    gradient Scan (mit-mot [0, 2, 1] -> [2, 1] + the in-loop weight-gradient Gemm) on the launch list."""
    x, h0, U = at.ftensor3("x"), at.ftensor3("h0"), at.fmatrix("U")

    def step(x_t, h2, h1, U):
        return at.tanh(at.dot(h1, U) + np.float32(0.5) * h2 + x_t)
    hs, _ = ae.scan(step, sequences=[x], outputs_info=[dict(initial=h0, taps=[-2, -1])], non_sequences=[U])
    cost = (hs * hs).sum()
    return [x, h0, U], [cost] + list(ae.grad(cost, [x, h0, U])), \
        [N((9, 16, 64), "float32", 4, 0.5), N((2, 16, 64), "float32", 3, 0.5), N((64, 64), "float32", 5, 0.125)]


@case("sm_taps13_b32_f32", rtol=1e-5, atol=1e-5)
def _():
    """Taps [-1, -3] and a second state with the usual single tap, width 96 (not a multiple of
    64: zero-padded weights), batch 32 (two batch blocks): the depth-2 value nobody names still
    needs its register."""
    x, h0, c0, U, V = at.ftensor3("x"), at.ftensor3("h0"), at.fmatrix("c0"), at.fmatrix("U"), at.fmatrix("V")

    def step(x_t, h3, h1, c, U, V):
        hn = at.tanh(at.dot(h1, U) + np.float32(0.25) * h3 * c + x_t)
        cn = at.sigmoid(at.dot(c, V) + hn)
        return hn, cn
    (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[dict(initial=h0, taps=[-3, -1]), c0],
                          non_sequences=[U, V])
    return [x, h0, c0, U, V], [hs, cs[-1]], \
        [N((7, 32, 96), "float32", 4, 0.5), N((3, 32, 96), "float32", 3, 0.5), N((32, 96), "float32", 2, 0.5),
         N((96, 96), "float32", 5, 0.1), N((96, 96), "float32", 6, 0.1)]



# ---------------------------------------------------------------------------------------
# recurrent-vector Scans of the class the persistent one-kernel loop covers
# (aesara_amd/scan_persist.py): circular output buffers (only the last state is read), two
# states (LSTM), a nit-sot projection of the new state + invariant vectors, a state length
# that is not a multiple of the wavefront; sizes modelled on tests/scan/test_basic.py
# ---------------------------------------------------------------------------------------
@case("sp_gru_last_f32", rtol=1e-5, atol=1e-5)
def _():
    x, h0 = at.fmatrix("x"), at.fvector("h0")
    Ws = [at.fmatrix(n) for n in ("Wz", "Uz", "Wr", "Ur", "Wh", "Uh")]

    def step(x_t, h, Wz, Uz, Wr, Ur, Wh, Uh):
        z = at.sigmoid(at.dot(x_t, Wz) + at.dot(h, Uz))
        r = at.sigmoid(at.dot(x_t, Wr) + at.dot(h, Ur))
        hh = at.tanh(at.dot(x_t, Wh) + at.dot(r * h, Uh))
        return (1 - z) * h + z * hh
    hs, _ = ae.scan(step, sequences=[x], outputs_info=[h0], non_sequences=Ws)
    H = 52
    return [x, h0] + Ws, [hs[-1]], \
        [N((21, H), "float32", 4, 0.3), N((H,), "float32", 3, 0.5)] + \
        [N((H, H), "float32", 5 + k, 1.0 / np.sqrt(H)) for k in range(6)]


@case("sp_lstm_vec_f32", rtol=2e-5, atol=2e-5)
def _():
    x, h0, c0 = at.fmatrix("x"), at.fvector("h0"), at.fvector("c0")
    names = ("Wi", "Ui", "Wf", "Uf", "Wo", "Uo", "Wg", "Ug")
    Ws = [at.fmatrix(n) for n in names]
    bf = at.fvector("bf")

    def step(x_t, h, c, Wi, Ui, Wf, Uf, Wo, Uo, Wg, Ug, bf):
        i = at.sigmoid(at.dot(x_t, Wi) + at.dot(h, Ui))
        f = at.sigmoid(at.dot(x_t, Wf) + at.dot(h, Uf) + bf)
        o = at.sigmoid(at.dot(x_t, Wo) + at.dot(h, Uo))
        g = at.tanh(at.dot(x_t, Wg) + at.dot(h, Ug))
        c2 = f * c + i * g
        return o * at.tanh(c2), c2
    (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[h0, c0], non_sequences=Ws + [bf])
    H, D = 48, 20
    shapes = [(D, H), (H, H)] * 4
    return [x, h0, c0] + Ws + [bf], [hs, cs[-1]], \
        [N((17, D), "float32", 4, 0.5), N((H,), "float32", 2, 0.3), N((H,), "float32", 3, 0.3)] + \
        [N(sh, "float32", 5 + k, 1.0 / np.sqrt(sh[0])) for k, sh in enumerate(shapes)] + \
        [N((H,), "float32", 20, 0.5, 1.0)]


@case("sp_rnn_proj_f32", rtol=1e-5, atol=1e-5)
def _():
    x, h0 = at.fmatrix("x"), at.fvector("h0")
    W, U_, V, b, c = at.fmatrix("W"), at.fmatrix("U"), at.fmatrix("V"), at.fvector("b"), at.fvector("c")

    def step(x_t, h, W, U_, V, b, c):
        h2 = at.tanh(at.dot(x_t, W) + at.dot(h, U_) + b)
        return h2, at.dot(h2, V) + c
    (hs, ys), _ = ae.scan(step, sequences=[x], outputs_info=[h0, None], non_sequences=[W, U_, V, b, c])
    H, D = 32, 12
    return [x, h0, W, U_, V, b, c], [hs, ys], \
        [N((19, D), "float32", 4, 0.5), N((H,), "float32", 2, 0.3), N((D, H), "float32", 5, 0.3),
         N((H, H), "float32", 6, 0.18), N((H, H), "float32", 7, 0.18), N((H,), "float32", 8, 0.1),
         N((H,), "float32", 9, 0.1)]


@case("sp_rnn_proj_narrow_f32", rtol=1e-5, atol=1e-5)
def _():
    # the projection is narrower than the state: outside the persistent class (launch-list path)
    x, h0 = at.fmatrix("x"), at.fvector("h0")
    W, U_, V = at.fmatrix("W"), at.fmatrix("U"), at.fmatrix("V")

    def step(x_t, h, W, U_, V):
        h2 = at.tanh(at.dot(x_t, W) + at.dot(h, U_))
        return h2, at.dot(h2, V)
    (hs, ys), _ = ae.scan(step, sequences=[x], outputs_info=[h0, None], non_sequences=[W, U_, V])
    H, D, P = 32, 12, 8
    return [x, h0, W, U_, V], [hs[-1], ys], \
        [N((15, D), "float32", 4, 0.5), N((H,), "float32", 2, 0.3), N((D, H), "float32", 5, 0.3),
         N((H, H), "float32", 6, 0.18), N((H, P), "float32", 7, 0.18)]


# ---------------------------------------------------------------------------------------
# Softmax family and Argmax (tests/tensor/test_special.py TestSoftmax/TestLogSoftmax/
# TestSoftmaxGrad; tests/tensor/test_math.py TestMaxAndArgmax)
# ---------------------------------------------------------------------------------------
for _dt, _tol in (("float64", 1e-12), ("float32", 3e-5)):
    def _mk(dt=_dt):
        from aesara.tensor.special import log_softmax, softmax
        x, t3 = T(dt, (2, 2), "x"), T(dt, (2, 2, 2), "t3")
        return [x, t3], [softmax(x, axis=-1), softmax(x, axis=0), softmax(t3, axis=1),
                         log_softmax(x, axis=-1), log_softmax(t3, axis=0),
                         at.log(softmax(x, axis=1))], \
            [N((9, 40), dt, 1, 3.0), N((4, 5, 6), dt, 2, 2.0)]
    case(f"softmax_family_{_dt}", rtol=_tol, atol=_tol)(_mk)

    def _mkn(dt=_dt):
        # axis=None: the reference's C code needs np.MAXDIMS (gone in NumPy 2) -> Python linker
        from aesara.tensor.special import log_softmax, softmax
        t3 = T(dt, (2, 2, 2), "t3")
        return [t3], [softmax(t3, axis=None), log_softmax(t3, axis=None)], \
            [N((4, 5, 6), dt, 2, 2.0)]
    case(f"softmax_axis_none_{_dt}", rtol=_tol, atol=_tol, ref_py=True)(_mkn)

    def _mkg(dt=_dt):
        from aesara.tensor.special import log_softmax, softmax
        x, w = T(dt, (2, 2), "x"), T(dt, (2, 2), "w")
        c1 = (softmax(x, axis=-1) * w).sum()
        c2 = (log_softmax(x, axis=0) * w).sum()
        return [x, w], [ae.grad(c1, x), c1, ae.grad(c2, x)], \
            [N((6, 17), dt, 3, 2.0), N((6, 17), dt, 4)]
    case(f"softmax_grad_{_dt}", rtol=_tol * 3, atol=_tol * 3)(_mkg)


@case("softmax_rows_f32", rtol=3e-5, atol=3e-6)
def _():
    from aesara.tensor.special import softmax
    x = at.fmatrix("x")
    return [x], [softmax(x, axis=-1)], [N((64, 1000), "float32", 1, 3.0)]


@case("logsoftmax_rows_f64", rtol=1e-12, atol=1e-12)
def _():
    from aesara.tensor.special import log_softmax
    x = at.dmatrix("x")
    return [x], [log_softmax(x, axis=-1)], [N((33, 77), "float64", 2, 3.0)]


for _dt, _tol in (("float64", 1e-11), ("float32", 5e-5)):
    def _mkln(dt=_dt):
        # a hand-written layer normalisation: CAReduce / DimShuffle / Elemwise nodes only
        x, g, b = T(dt, (2, 2, 2), "x"), T(dt, (2,), "g"), T(dt, (2,), "b")
        mu = x.mean(axis=-1, keepdims=True)
        var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
        y = (x - mu) / at.sqrt(var + np.asarray(1e-5, dt)) * g + b
        return [x, g, b], [y, mu, var[..., 0]], \
            [N((3, 7, 96), dt, 1, 2.0, 0.5), N((96,), dt, 2), N((96,), dt, 3)]
    case(f"layernorm_{_dt}", rtol=_tol, atol=_tol)(_mkln)


@case("softmax_masked_scaled_f32", rtol=3e-5, atol=3e-6)
def _():
    # attention-style: softmax(scores / temperature + mask) — the producer Elemwise joins the
    # row chain, the scaled logits never reach memory
    from aesara.tensor.special import softmax
    sc, mask, tau = at.ftensor3("s"), at.fmatrix("mask"), at.fscalar("tau")
    return [sc, mask, tau], [softmax(sc / tau + mask.dimshuffle("x", 0, 1), axis=-1)], \
        [N((3, 17, 48), "float32", 1, 2.0), U((17, 48), "float32", 2, -4.0, 0.0),
         K(0.7, "float32")]


@case("split_and_join_grad", rtol=1e-12, atol=1e-12)
def _():
    # Split itself, and the gradient of a concatenation (Join.grad emits Split)
    x, y, sp = at.dmatrix("x"), at.dmatrix("y"), at.lvector("sp")
    parts = at.split(x, sp, 3, axis=1)
    c = at.concatenate([x, y], axis=1)
    cost = (c ** 2 * at.arange(c.shape[1])).sum()
    return [x, y, sp], [parts[0], parts[1] * 2.0, parts[2]] + list(ae.grad(cost, [x, y])), \
        [N((5, 9), seed=1), N((5, 4), seed=2), {"kind": "const_list", "values": [2, 3, 4],
                                                "shape": [3], "dtype": "int64"}]


@case("cumop_int", exact=True, ref_py=True)
def _():
    # int64 only: for narrower ints the reference's Python perform returns NumPy's promoted int64
    # although the Op declares the input dtype (its C code keeps it) — not a behaviour to pin
    x, t = at.lmatrix("x"), at.tensor3("t", dtype="int64")
    return [x, t], [at.cumsum(x, axis=0), at.cumsum(x, axis=1), at.cumsum(x), at.cumprod(x % 3 + 1, axis=1),
                    at.cumsum(t, axis=1), at.cumsum(t.dimshuffle(2, 0, 1), axis=2)], \
        [I((7, 150), "int64", 1, -9, 9), I((3, 70, 5), "int64", 2, -99, 99)]


@case("cumop_float", rtol=1e-5, atol=1e-5, ref_py=True)
def _():
    x, v = at.fmatrix("x"), at.dvector("v")
    return [x, v], [at.cumsum(x, axis=1), at.cumsum(x, axis=0), at.cumsum(v), at.cumprod(v * 0.1 + 1.0)], \
        [N((6, 200), "float32", 1), N((300,), "float64", 2)]


for _dt, _tol in (("float64", 1e-11), ("float32", 3e-5)):
    def _mkmlp(dt=_dt):
        # small-batch layers: products + bias + activation (fused GEMM-chain epilogue kernels),
        # both weight layouts, two products feeding one Elemwise, ragged sizes
        x, y = T(dt, (2, 2), "x"), T(dt, (2, 2), "y")
        W, W2, U_, Wt, b = (T(dt, (2, 2), n) for n in ("W", "W2", "U", "Wt", "b2"))
        b = T(dt, (2,), "b")
        h1 = at.tanh(at.dot(x, W) + b)
        h2 = at.sigmoid(at.dot(x, W2) + at.dot(y, U_)) * b
        h3 = at.dot(x, Wt.T) * 2.0 + b
        h4 = at.maximum(at.dot(y, U_.T[::1].T * 1.0) + at.dot(x, (Wt * 0.5).T), 0.0)
        return [x, y, W, W2, U_, Wt, b], [h1, h2, h3, h4], \
            [N((37, 40), dt, 1), N((37, 24), dt, 2), N((40, 52), dt, 3, 0.3),
             N((40, 52), dt, 7, 0.3), N((24, 52), dt, 4, 0.3), N((52, 40), dt, 5, 0.3),
             N((52,), dt, 6)]
    case(f"mlp_layers_{_dt}", rtol=_tol, atol=_tol)(_mkmlp)


@case("empty_inputs", exact=True)
def _():
    # zero-size operands (tests/tensor/test_elemwise.py TestCAReduce cases with 0 extents,
    # test_subtensor.py empty index vectors)
    from aesara.tensor.special import softmax
    x, y, v, i = at.dmatrix("x"), at.dmatrix("y"), at.dvector("v"), at.lvector("i")
    return [x, y, v, i], [x + y, at.exp(x) * 2.0, x.sum(), x.sum(axis=0), x.sum(axis=1),
                          x.prod(axis=0), at.dot(x.T, y), v[i], v[i] * 2.0,
                          at.concatenate([x, y], axis=0), x.T + 1.0], \
        [N((0, 5), seed=1), N((0, 5), seed=2), N((7,), seed=3), I((0,), "int64", 4, 0, 7)]


for _dt, _tol in (("float64", 1e-10), ("float32", 5e-5)):
    def _mklstm(dt=_dt):
        # LSTM (gates from one concatenated product, sliced) forward + BPTT: loss and the three
        # parameter gradients (the backward Scan has matrix-valued mit-mot accumulators)
        H = 8
        x, y = T(dt, (2, 2, 2), "x"), T(dt, (2, 2), "y")
        W, U_, b = T(dt, (2, 2), "W"), T(dt, (2, 2), "U"), T(dt, (2,), "b")

        def step(x_t, h, c, W, U_, b):
            g = at.dot(x_t, W) + at.dot(h, U_) + b
            i, f, o, gg = (g[:, k * H:(k + 1) * H] for k in range(4))
            c2 = at.sigmoid(f) * c + at.sigmoid(i) * at.tanh(gg)
            return at.sigmoid(o) * at.tanh(c2), c2
        z = at.zeros((x.shape[1], H), dtype=dt)
        (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[z, z], non_sequences=[W, U_, b])
        loss = ((hs[-1] - y) ** 2).mean()
        return [x, y, W, U_, b], [loss] + list(ae.grad(loss, [W, U_, b])), \
            [N((5, 4, 6), dt, 1), N((4, 8), dt, 2), N((6, 32), dt, 3, 0.3), N((8, 32), dt, 4, 0.3),
             N((32,), dt, 5, 0.1)]
    case(f"lstm_bptt_{_dt}", rtol=_tol, atol=_tol)(_mklstm)


@case("lstm_fused_fwd_f32", rtol=2e-5, atol=2e-5)
def _():
    """The usual fused-gate LSTM step (ONE product for the four gates, sliced by columns) as a
    forward Scan with a matrix state, H = 64: all hidden states and the last cell state."""
    H = 64
    x, h0, c0 = at.ftensor3("x"), at.fmatrix("h0"), at.fmatrix("c0")
    W, U_, b = at.fmatrix("W"), at.fmatrix("U"), at.fvector("b")

    def step(x_t, h, c, W, U_, b):
        g = at.dot(x_t, W) + at.dot(h, U_) + b
        i, f, o, gg = (g[:, k * H:(k + 1) * H] for k in range(4))
        c2 = at.sigmoid(f) * c + at.sigmoid(i) * at.tanh(gg)
        return at.sigmoid(o) * at.tanh(c2), c2
    (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[h0, c0], non_sequences=[W, U_, b])
    return [x, h0, c0, W, U_, b], [hs, cs[-1]], \
        [N((6, 3, 20), "float32", 1, 0.5), N((3, H), "float32", 2, 0.5), N((3, H), "float32", 6, 0.5),
         N((20, 4 * H), "float32", 3, 0.2), N((H, 4 * H), "float32", 4, 0.12), N((4 * H,), "float32", 5, 0.1)]


@case("lstm_fused_vec_f32", rtol=2e-5, atol=2e-5)
def _():
    """The fused-gate LSTM step on a VECTOR state (one sequence, B = 1): ``dot(x_t, W) + dot(h, U)
    + b`` of length 4H sliced into the gates, H = 48."""
    H = 48
    x, h0, c0 = at.fmatrix("x"), at.fvector("h0"), at.fvector("c0")
    W, U_, b = at.fmatrix("W"), at.fmatrix("U"), at.fvector("b")

    def step(x_t, h, c, W, U_, b):
        g = at.dot(x_t, W) + at.dot(h, U_) + b
        i, f, o, gg = (g[k * H:(k + 1) * H] for k in range(4))
        c2 = at.sigmoid(f) * c + at.sigmoid(i) * at.tanh(gg)
        return at.sigmoid(o) * at.tanh(c2), c2
    (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[h0, c0], non_sequences=[W, U_, b])
    return [x, h0, c0, W, U_, b], [hs, cs[-1]], \
        [N((7, 20), "float32", 1, 0.5), N((H,), "float32", 2, 0.5), N((H,), "float32", 6, 0.5),
         N((20, 4 * H), "float32", 3, 0.2), N((H, 4 * H), "float32", 4, 0.12), N((4 * H,), "float32", 5, 0.1)]


@case("rnn_bias_bptt_b4_f32", rtol=3e-4, atol=3e-5)
def _():
    """A batched tanh RNN WITH a bias under aesara.grad: the bias gradient is a sum over the batch
    accumulated inside the gradient Scan (a sit-sot ``acc + sum(delta, axis=0)``)."""
    x, h0 = at.ftensor3("x"), at.fmatrix("h0")
    W, U_, b = at.fmatrix("W"), at.fmatrix("U"), at.fvector("b")

    def step(x_t, h, W, U_, b):
        return at.tanh(at.dot(x_t, W) + at.dot(h, U_) + b)
    hs, _ = ae.scan(step, sequences=[x], outputs_info=[h0], non_sequences=[W, U_, b])
    loss = (hs[-1] ** 2).sum() + hs.mean()
    H = 16
    return [x, h0, W, U_, b], [loss] + list(ae.grad(loss, [W, U_, b, h0])), \
        [N((7, 4, 12), "float32", 1, 0.5), N((4, H), "float32", 2, 0.5),
         N((12, H), "float32", 3, 0.3), N((H, H), "float32", 4, 0.25), N((H,), "float32", 5, 0.1)]


@case("lstm_fused_bptt_h64_f32", rtol=3e-4, atol=3e-5)
def _():
    """Fused-gate LSTM (matrix state, H = 64) under aesara.grad: loss and gradients wrt W, U, b
    (the step of lstm_fused_fwd_f32; a width the GPU probes can run at a realistic T and B)."""
    H = 64
    x, h0, c0 = at.ftensor3("x"), at.fmatrix("h0"), at.fmatrix("c0")
    W, U_, b = at.fmatrix("W"), at.fmatrix("U"), at.fvector("b")

    def step(x_t, h, c, W, U_, b):
        g = at.dot(x_t, W) + at.dot(h, U_) + b
        i, f, o, gg = (g[:, k * H:(k + 1) * H] for k in range(4))
        c2 = at.sigmoid(f) * c + at.sigmoid(i) * at.tanh(gg)
        return at.sigmoid(o) * at.tanh(c2), c2
    (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[h0, c0], non_sequences=[W, U_, b])
    loss = (hs[-1] ** 2).sum() + cs.mean()
    return [x, h0, c0, W, U_, b], [loss] + list(ae.grad(loss, [W, U_, b])), \
        [N((5, 3, 20), "float32", 1, 0.5), N((3, H), "float32", 2, 0.5), N((3, H), "float32", 6, 0.5),
         N((20, 4 * H), "float32", 3, 0.2), N((H, 4 * H), "float32", 4, 0.12), N((4 * H,), "float32", 5, 0.1)]


@case("lstm_fused_vec_bptt_f32", rtol=3e-4, atol=3e-5)
def _():
    """Fused-gate LSTM on a vector state under aesara.grad: loss and the gradients wrt W, U, b."""
    H = 16
    x, h0, c0 = at.fmatrix("x"), at.fvector("h0"), at.fvector("c0")
    W, U_, b = at.fmatrix("W"), at.fmatrix("U"), at.fvector("b")

    def step(x_t, h, c, W, U_, b):
        g = at.dot(x_t, W) + at.dot(h, U_) + b
        i, f, o, gg = (g[k * H:(k + 1) * H] for k in range(4))
        c2 = at.sigmoid(f) * c + at.sigmoid(i) * at.tanh(gg)
        return at.sigmoid(o) * at.tanh(c2), c2
    (hs, cs), _ = ae.scan(step, sequences=[x], outputs_info=[h0, c0], non_sequences=[W, U_, b])
    loss = (hs[-1] ** 2).sum() + cs.mean()
    return [x, h0, c0, W, U_, b], [loss] + list(ae.grad(loss, [W, U_, b])), \
        [N((8, 12), "float32", 1, 0.5), N((H,), "float32", 2, 0.5), N((H,), "float32", 6, 0.5),
         N((12, 4 * H), "float32", 3, 0.25), N((H, 4 * H), "float32", 4, 0.2), N((4 * H,), "float32", 5, 0.1)]


@case("rowchain_integer", exact=True)
def _():
    # last-axis reduction chains on integer / bool data go through the same row-chain kernels
    x, b = at.imatrix("x"), at.matrix("b", dtype="bool")
    t3 = at.tensor3("t3", dtype="int64")
    mx = x.max(axis=-1, keepdims=True)
    return [x, b, t3], [x - mx, (x - x.min(axis=1, keepdims=True)) * 2,
                        at.eq(x, mx).sum(axis=1), b & b.any(axis=1, keepdims=True),
                        t3 - t3.sum(axis=2, keepdims=True), (t3 % 7).prod(axis=-1, keepdims=True) + t3], \
        [I((37, 19), "int32", 1, -99, 99), B((11, 70), 2, 0.2), I((3, 5, 9), "int64", 3, -9, 9)]


@case("argmax_axes", exact=True)
def _():
    x, m, v = at.dtensor3("x"), at.imatrix("m"), at.dvector("v")
    b = at.matrix("b", dtype="bool")
    mx, am = at.max_and_argmax(m, axis=1)
    return [x, m, v, b], [at.argmax(x, axis=None), at.argmax(x, axis=0), at.argmax(x, axis=2),
                          at.argmax(x, axis=[0, 2]), at.argmax(x.dimshuffle(2, 0, 1), axis=1),
                          mx, am, at.argmax(v), at.argmax(b, axis=0), at.argmax(m.T, axis=1)], \
        [I((5, 6, 7), "float64", 1, -3, 3), I((9, 130), "int32", 2, -50, 50),
         {"kind": "normal_with_nan", "shape": [301], "dtype": "float64", "seed": 7},
         B((5, 9), 3, 0.3)]

# ---------------------------------------------------------------------------------------
# ARange and N-d integer-array indexing (tests/tensor/test_subtensor.py TestAdvancedSubtensor,
# test_basic.py TestARange); the NLL idiom log_softmax(x)[arange(n), y] and its gradient
# ---------------------------------------------------------------------------------------
@case("arange_variants", exact=True)
def _():
    n, a, b = at.lscalar("n"), at.lscalar("a"), at.lscalar("b")
    return [n, a, b], [at.arange(n), at.arange(a, n, b), at.arange(n, a, -b),
                       at.arange(n, dtype="int32") * 2, at.arange(a, a)], \
        [K(11, "int64"), K(2, "int64"), K(3, "int64")]


@case("arange_float", rtol=1e-6, atol=1e-6)
def _():
    s, e = at.dscalar("s"), at.dscalar("e")
    return [s, e], [at.arange(s, e, 0.25), at.arange(s, e, 0.1, dtype="float32")], \
        [K(-1.5, "float64"), K(3.2, "float64")]


@case("advsub_nd", exact=True)
def _():
    x, t = at.imatrix("x"), at.tensor3("t", dtype="int32")
    i, j = at.lvector("i"), at.ivector("j")
    ci = at.lmatrix("ci")
    return [x, t, i, j, ci], [x[i, j], x[at.arange(i.shape[0]), j], t[i, j], t[i, j, i],
                              x[ci, j.dimshuffle("x", 0)], x.T[j, i]], \
        [I((9, 7), "int32", 1, -99, 99), I((9, 7, 9), "int32", 2, -99, 99),
         I((6,), "int64", 3, -9, 9), I((6,), "int32", 4, -7, 7), I((4, 1), "int64", 5, 0, 9)]


@case("advincsub_nd", exact=True, ref_py=True)
def _():
    x, y = at.imatrix("x"), at.ivector("y")
    i, j = at.lvector("i"), at.ivector("j")
    from aesara.tensor.subtensor import inc_subtensor, set_subtensor
    return [x, y, i, j], [inc_subtensor(x[i, j], y), inc_subtensor(x[i, j], 5),
                          set_subtensor(x[i, i], y)], \
        [I((9, 7), "int32", 1, -99, 99), I((6,), "int32", 2, -9, 9),
         {"kind": "perm", "n": 7, "shape": [6], "dtype": "int64", "seed": 3},
         I((6,), "int32", 4, -7, 7)]


for _dt, _tol in (("float64", 1e-12), ("float32", 3e-5)):
    def _mknll(dt=_dt):
        from aesara.tensor.special import log_softmax
        x, W, b, y = T(dt, (2, 2), "x"), T(dt, (2, 2), "W"), T(dt, (2,), "b"), at.lvector("y")
        logits = at.dot(x, W) + b
        nll = -log_softmax(logits, axis=-1)[at.arange(y.shape[0]), y].mean()
        gW, gb = ae.grad(nll, [W, b])
        return [x, W, b, y], [nll, gW, gb, at.argmax(logits, axis=1)], \
            [N((48, 20), dt, 1), N((20, 10), dt, 2, 0.5), N((10,), dt, 3, 0.1),
             I((48,), "int64", 4, 0, 10)]
    case(f"nll_classifier_{_dt}", rtol=_tol, atol=_tol)(_mknll)

# ---------------------------------------------------------------------------------------
# BASELINE.json configs at reduced shapes
# ---------------------------------------------------------------------------------------
@case("cfg1a_scalar_add", exact=True)
def _():
    a, b = at.dscalar("a"), at.dscalar("b")
    return [a, b], [a + b], [K(1.5, "float64"), K(2.5, "float64")]


@case("cfg1b_matrix_add", exact=True)
def _():
    x, y = at.dmatrix("x"), at.dmatrix("y")
    return [x, y], [x + y], [U((128, 96), seed=0), U((128, 96), seed=1)]


@case("cfg2_gauss_sum", rtol=1e-12, atol=0)
def _():
    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    return [x, mu, sg], [at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum()], \
        [N((256, 192), seed=1), K(0.1, "float64"), K(1.3, "float64")]


@case("cfg3a_gemv", rtol=1e-12, atol=1e-11)
def _():
    M, v, a = at.dmatrix("M"), at.dvector("v"), at.dscalar("a")
    return [M, v, a], [a / a + at.dot(M + a, v), at.dot(M, v) + a], \
        [N((200, 300), seed=2), N((300,), seed=3), K(2.0, "float64")]


@case("cfg3b_gemm_update", rtol=2e-5, atol=2e-4)
def _():
    Cc, A, Bm = at.fmatrix("C"), at.fmatrix("A"), at.fmatrix("B")
    return [Cc, A, Bm], [np.float32(0.4) * Cc + np.float32(0.8) * at.dot(A, Bm)], \
        [N((256, 256), "float32", 1), N((256, 256), "float32", 3), N((256, 256), "float32", 4)]


@case("cfg5_logistic", rtol=2e-5, atol=1e-3)
def _():
    X, w, b, y = at.fmatrix("X"), at.fvector("w"), at.fscalar("b"), at.fvector("y")
    p = at.sigmoid(at.dot(X, w) + b)
    logp = (y * at.log(p) + (1 - y) * at.log(1 - p)).sum()
    gw, gb = ae.grad(logp, [w, b])
    return [X, w, b, y], [logp, gw, gb], \
        [N((4096, 256), "float32", 6), N((256,), "float32", 7, 1.0 / 16), K(0.1, "float32"),
         B((4096,), 8, 0.5, "float32")]


# ---------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------
# round 6
# ---------------------------------------------------------------------------------------
@case("ew_exact_quotient_floor", exact=True)
def _():
    # a RUN-TIME scalar divisor must give the IEEE quotient (the reference divides; only
    # constant divisors are folded into reciprocals by its canonicaliser): (7 k) / 7 == k exactly, so
    # floor / eq / integer casts behind the division see integers (binning: floor(x / width)).
    # Every quotient below reaches memory element-wise -> the correctly rounded path, bit for bit.
    x, c, xd, w = at.fmatrix("x"), at.fscalar("c"), at.dmatrix("xd"), at.dscalar("w")
    q32, q64 = x / c, xd / w
    return [x, c, xd, w], [at.floor(q32), q32, at.cast(q32, "int32"), at.floor(q64), q64,
                           at.cast(at.floor((-0.5 * xd) / w), "int64"), (2.0 * at.sqr(xd)) / w,
                           # a COMPUTED loop-invariant divisor: the hoisted-reciprocal + Markstein path
                           at.floor(xd / at.sqr(w)), (2.0 * xd) / at.sqr(w), (-0.5 * x) / (c * c + c)], \
        [{"kind": "arange", "shape": [64, 33], "dtype": "float32", "scale": 7}, K(7.0, "float32"),
         {"kind": "arange", "shape": [64, 33], "dtype": "float64", "scale": 1.3}, K(1.3, "float64")]


from aesara.tensor import extra_ops as _xo  # noqa: E402


@case("repeat_scalar_and_vector", exact=True)
def _():
    # tensor/extra_ops.py:637 Repeat: scalar repeats (broadcast copy), vector repeats (searchsorted
    # over the running sum), along an inner axis, flattened, and a length-1 repeats vector
    x, r0, rv, m = at.dmatrix("x"), at.iscalar("r"), at.lvector("rv"), at.imatrix("m")
    r1 = at.lvector("r1")
    return [x, r0, rv, m, r1], [
        _xo.repeat(x, r0, axis=0), _xo.repeat(x, r0, axis=1), _xo.repeat(x, r0),
        _xo.repeat(x, rv, axis=0), _xo.repeat(m, rv, axis=0), _xo.repeat(x.T, rv, axis=1),
        _xo.Repeat(axis=1)(x, r1)], \
        [N((5, 7), seed=11), K(3, "int32"), {"kind": "const_list", "shape": [5], "dtype": "int64",
                                             "values": [2, 0, 1, 4, 3]},
         I((5, 3), "int32", 12, -50, 50), {"kind": "const_list", "shape": [1], "dtype": "int64", "values": [2]}]


@case("searchsorted_sides_sorter_nan", exact=True)
def _():
    # tensor/extra_ops.py:102 SearchsortedOp: both sides, a sorter, NaN keys (above every number),
    # mixed dtypes (int keys in a float sequence)
    x, v, xi, vi, s = at.dvector("x"), at.dmatrix("v"), at.lvector("xi"), at.ivector("vi"), at.lvector("s")
    xs = at.sort(x)
    return [x, v, xi, vi, s], [
        _xo.searchsorted(xs, v), _xo.searchsorted(xs, v, side="right"),
        _xo.searchsorted(at.sort(xi), vi), _xo.searchsorted(at.sort(xi), vi, side="right"),
        _xo.searchsorted(xs, vi), _xo.searchsorted(x, v, sorter=at.argsort(x))], \
        [{"kind": "normal_with_nan", "seed": 3, "shape": [64], "dtype": "float64"},
         {"kind": "normal_with_nan", "seed": 4, "shape": [6, 9], "dtype": "float64"},
         I((40,), "int64", 5, -8, 8), I((25,), "int32", 6, -10, 10), I((3,), "int64", 7, 0, 2)]


@case("unique_all_returns", exact=True)
def _():
    # tensor/extra_ops.py:1152 Unique on vectors: values / first indices / inverse / counts, with
    # repeated values and several NaNs (one run)
    x, k = at.dvector("x"), at.ivector("k")
    outs = list(_xo.Unique(True, True, True)(x)) + list(_xo.Unique(False, True, True)(k)) + \
        [_xo.unique(k), _xo.Unique(True, False, False)(k)[1]]
    return [x, k], outs, [{"kind": "normal_with_nan", "seed": 8, "shape": [200], "dtype": "float64"},
                          I((300,), "int32", 9, -20, 20)]


@case("topk_values_and_indices", exact=True)
def _():
    # tensor/sort.py:309 TopKOp: order of the result is not specified by the reference (np.partition)
    # -> compared after a sort; distinct keys so the index sets are determined
    from aesara.tensor.sort import argtopk, topk
    x, m, k = at.dvector("x"), at.fmatrix("m"), at.iscalar("k")
    return [x, m, k], [
        at.sort(topk(x, k, sorted=False)), at.sort(argtopk(x, k, sorted=False)),
        at.sort(topk(x, -k, sorted=False)), at.sort(argtopk(x, -k, sorted=False, idx_dtype="int32")),
        at.sort(topk(m, k, axis=0, sorted=False), axis=0), at.sort(argtopk(m, 3, axis=1, sorted=False), axis=1)], \
        [{"kind": "perm", "n": 500, "shape": [257], "dtype": "float64"},
         {"kind": "perm", "n": 4000, "shape": [40, 33], "dtype": "float32"}, K(7, "int32")]


@case("ravel_unravel_index", exact=True)
def _():
    # tensor/extra_ops.py:1362 RavelMultiIndex (raise / wrap / clip, C / F) and :1283 UnravelIndex
    i, j, k = at.lmatrix("i"), at.lmatrix("j"), at.lmatrix("k")
    flat = at.lvector("flat")
    dims = (4, 5, 6)
    outs = [_xo.ravel_multi_index((i % 4, j % 5, k % 6), dims),
            _xo.ravel_multi_index((i, j, k), dims, mode="wrap"),
            _xo.ravel_multi_index((i, j, k), dims, mode="clip", order="F")]
    outs += list(_xo.unravel_index(flat, dims)) + list(_xo.unravel_index(flat, dims, order="F"))
    return [i, j, k, flat], outs, [I((7, 3), "int64", 1, -20, 20), I((7, 3), "int64", 2, -20, 20),
                                   I((7, 3), "int64", 3, -20, 20), I((50,), "int64", 4, 0, 120)]


@case("choose_permute_rows", exact=True)
def _():
    # tensor/basic.py:3773 Choose (raise / wrap / clip, broadcasting of a against the choices) and
    # :3111 PermuteRowElements (one permutation for all rows, one per row, inverse)
    from aesara.tensor.basic import choose, inverse_permutation, permute_row_elements
    a, a1 = at.imatrix("a"), at.ivector("a1")
    ch = at.dtensor3("ch")
    x, p1, p2 = at.dmatrix("x"), at.lvector("p1"), at.lmatrix("p2")
    return [a, a1, ch, x, p1, p2], [
        choose(a % 4, ch), choose(a, ch, mode="wrap"), choose(a, ch, mode="clip"), choose(a1 % 4, ch),
        permute_row_elements(x, p1), permute_row_elements(x, p2), permute_row_elements(x, p2, True),
        inverse_permutation(p2)], \
        [I((5, 6), "int32", 1, -9, 9), I((6,), "int32", 2, -9, 9), N((4, 5, 6), seed=3),
         N((5, 6), seed=4), {"kind": "perm", "n": 6, "shape": [6], "dtype": "int64"},
         {"kind": "const_list", "shape": [5, 6], "dtype": "int64",
          "values": [[3, 1, 0, 5, 4, 2], [0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [1, 0, 3, 2, 5, 4],
                     [2, 3, 4, 5, 0, 1]]}]


@case("bartlett_filldiag_offset_contiguous", rtol=1e-13, atol=1e-15)
def _():
    # tensor/extra_ops.py:822 Bartlett, :980 FillDiagonalOffset (wide / tall, both signs), :40 CpuContiguous
    M, M1 = at.iscalar("M"), at.iscalar("M1")
    a, t, v, o1, o2 = at.dmatrix("a"), at.dmatrix("t"), at.dscalar("v"), at.iscalar("o1"), at.iscalar("o2")
    return [M, M1, a, t, v, o1, o2], [
        _xo.bartlett(M), _xo.bartlett(M1), _xo.fill_diagonal_offset(a, v, o1),
        _xo.fill_diagonal_offset(a, v, o2), _xo.fill_diagonal_offset(t, v, o1),
        _xo.fill_diagonal_offset(t, v, o2), _xo.cpu_contiguous(a.T) * 2.0], \
        [K(12, "int32"), K(1, "int32"), N((5, 9), seed=1), N((9, 4), seed=2), K(-7.5, "float64"),
         K(2, "int32"), K(-3, "int32")]


for _dt in ("float64", "float32"):
    def _mkxent(dt=_dt):
        # tensor/nnet/basic.py:57 SoftmaxWithBias, :458 CrossentropySoftmaxArgmax1HotWithBias,
        # :716 CrossentropySoftmax1HotWithBiasDx, :1655 / :1707 Prepend_scalar*_to_each_row
        from aesara.tensor.nnet.basic import (crossentropy_softmax_1hot_with_bias_dx,
                                              crossentropy_softmax_argmax_1hot_with_bias,
                                              prepend_0_to_each_row, prepend_scalar_to_each_row,
                                              softmax_with_bias)
        x, b, y, dy = T(dt, (8, 8), "x"), T(dt, (8,), "b"), at.lvector("y"), T(dt, (8,), "dy")
        nll, sm, am = crossentropy_softmax_argmax_1hot_with_bias(x, b, y)
        return [x, b, y, dy], [softmax_with_bias(x, b), nll, sm, am,
                               crossentropy_softmax_1hot_with_bias_dx(dy, sm, y),
                               prepend_0_to_each_row(x), prepend_scalar_to_each_row(b[0], x)], \
            [N((33, 10), dt, 1, 2.0), N((10,), dt, 2), I((33,), "int64", 3, 0, 10), N((33,), dt, 4)]
    case(f"nnet_row_programs_{_dt}", rtol=2e-5 if _dt == "float32" else 1e-12,
         atol=1e-6 if _dt == "float32" else 1e-13)(_mkxent)


@case("op_from_graph_inlined", rtol=1e-12, atol=1e-13)
def _():
    # compile/builders.py:188 OpFromGraph: the inner graph is expanded at lowering
    from aesara.compile.builders import OpFromGraph
    x, y, z = at.dmatrices("xyz")
    ofg = OpFromGraph([x, y, z], [at.exp(x) * y + z, (x + y).sum(axis=0)])
    a, b, c = at.dmatrix("a"), at.dmatrix("b"), at.dmatrix("c")
    o1, o2 = ofg(a, b, c)
    return [a, b, c], [o1 * 2.0, o2, ofg(b, a, c)[0]], [N((6, 7), seed=1), N((6, 7), seed=2), N((6, 7), seed=3)]


@case("xlogx_xlogy0", rtol=1e-13, atol=0)
def _():
    # tensor/xlogx.py:7 XlogX / :36 XlogY0 (0 * log(0) = 0)
    from aesara.tensor.xlogx import xlogx, xlogy0
    x, y = at.dmatrix("x"), at.dmatrix("y")
    xz = at.switch(x < 0.3, 0.0, x)
    return [x, y], [xlogx(xz), xlogy0(xz, y), xlogy0(xz, at.zeros_like(y))], [U((7, 9), seed=1), U((7, 9), seed=2)]


def _ultra_fast(dtype, n, seed):
    # tensor/nnet/sigm.py:20 UltraFastScalarSigmoid (c_code :54): the three pieces of the tanh
    # approximation on both signs (variables of the output type, double arithmetic between them).
    # Not bit-exact by construction: the reference's own value of a + b * (x - 1.7) depends on whether
    # its compiler contracts it into an fma (g++ -O3 -march=native does, NumPy does not)
    from aesara.tensor.nnet.sigm import ultra_fast_sigmoid
    x = at.vector("x", dtype=dtype)
    return [x], [ultra_fast_sigmoid(x), ultra_fast_sigmoid(-x * 0.5) * 2], [N((n,), dtype, seed=seed, scale=4.0)]


@case("ultra_fast_sigmoid_f64", rtol=1e-15, atol=2.3e-16)
def _():
    return _ultra_fast("float64", 257, 1)


@case("ultra_fast_sigmoid_f32", rtol=2e-7, atol=1.2e-7)
def _():
    return _ultra_fast("float32", 300, 2)


@case("incomplete_gamma_family", rtol=1e-9, atol=1e-12)
def _():
    # scalar/math.py:580 GammaInc / :629 GammaIncC / :538 Chi2SF / :836 GammaU / :877 GammaL (the
    # reference's C bodies: c_code/gamma.c series / continued fraction).  GammaU's C body is the
    # continued fraction WHATEVER x is: for x < k + 1 it has not converged after its 1024 steps and
    # the reference's own value depends on how its compiler contracts a*d + b (0.2 % off SciPy at
    # k = 10, x = 0.25) — GammaU is pinned where the fraction converges (x2 >= k + 1)
    k, x, x2 = at.dmatrix("k"), at.dmatrix("x"), at.dmatrix("x2")
    return [k, x, x2], [at.gammainc(k, x), at.gammaincc(k, x), at.chi2sf(x, k), at.gammau(k, x2),
                        at.gammal(k, x)], \
        [U((9, 11), seed=1, low=0.3, high=12.0), U((9, 11), seed=2, low=0.0, high=25.0),
         U((9, 11), seed=3, low=13.5, high=40.0)]


@case("inplace_transfer_dtype", exact=True)
def _():
    # scalar/basic.py transfer_type: add_inplace(int64, float64) computes the float sum, then truncates
    # into the int64 output (Elemwise inplace Ops built by the caller, tensor/inplace.py)
    from aesara.compile.ops import deep_copy_op
    from aesara.tensor import inplace
    x, y = at.lmatrix("x"), at.dmatrix("y")
    return [x, y], [inplace.add_inplace(deep_copy_op(x), y), inplace.mul_inplace(deep_copy_op(x), y)], \
        [I((5, 7), "int64", 1, -9, 9), N((5, 7), seed=2)]


@case("sort_rows_longer_than_lds", exact=True)
def _():
    # tensor/sort.py:29 SortOp / :150 ArgSortOp on rows that do not fit a workgroup's LDS (chunk sort +
    # merge passes, csrc/sort.hip): NaNs last, ties in input order (stable), ragged lengths, several
    # rows, an int8 row with many ties, and the Ops built on the sort (Unique, TopK) at that size
    from aesara.tensor.sort import argtopk, topk
    x, m, k8, ki = at.dvector("x"), at.fmatrix("m"), at.bvector("k8"), at.ivector("ki")
    return [x, m, k8, ki], [
        at.sort(x), at.argsort(x, kind="stable"), at.sort(m, axis=1), at.argsort(m, axis=0, kind="stable"),
        at.sort(k8), at.argsort(k8, kind="stable"),
        _xo.Unique(True, True, True)(ki)[0], _xo.Unique(True, True, True)(ki)[1],
        _xo.Unique(True, True, True)(ki)[2], _xo.Unique(True, True, True)(ki)[3],
        at.sort(topk(x[:20000], 777, sorted=False))], \
        [{"kind": "normal_with_nan", "seed": 11, "shape": [33333], "dtype": "float64"},
         N((3, 20011), "float32", 12), I((70001,), "int8", 13, -128, 127), I((50000,), "int32", 14, -3000, 3000)]


@case("unique_rows_as_items", exact=True)
def _():
    # tensor/extra_ops.py:1152 Unique with an axis: the slices along it are the items (np.unique(x, axis=k):
    # lexicographic order of the rows / columns), all four outputs, a 3-d input along its middle axis
    m, t = at.imatrix("m"), at.ltensor3("t")
    u0 = _xo.Unique(True, True, True, axis=0)(m)
    u1 = _xo.Unique(True, True, True, axis=1)(m)
    return [m, t], list(u0) + list(u1) + [_xo.Unique(False, False, False, axis=1)(t)], \
        [I((40, 3), "int32", 21, 0, 3), I((5, 17, 2), "int64", 22, 0, 2)]


def _close(a, b, exact, rtol, atol):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False, f"shape/dtype {a.shape}/{a.dtype} vs {b.shape}/{b.dtype}"
    if exact:
        ok = np.array_equal(a, b, equal_nan=True)
        return ok, "exact mismatch" if not ok else ""
    ok = np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
    if not ok:
        with np.errstate(all="ignore"):
            err = np.nanmax(np.abs(a.astype("float64") - b.astype("float64")))
        return False, f"max abs err {err}"
    return True, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="*")
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    cases_path = os.path.join(GOLDEN, "cases.json")
    index = {}
    if os.path.exists(cases_path) and args.only != "*":
        with open(cases_path) as f:
            index = {c["name"]: c for c in json.load(f)["cases"]}
    n_ok = 0
    for name, fn, exact, rtol, atol in CASES:
        if not fnmatch.fnmatch(name, args.only):
            continue
        print(f"... {name}", flush=True)
        ins, outs, specs = fn()
        xs = [make_input(s) for s in specs]
        # 2. reference's own linker (C thunks under the CVM)
        # (REF_PY: no C compiler for this case at all — the inner function of a Scan compiles with the
        # default mode whatever the outer linker is, and e.g. AdvancedIncSubtensor1's C body does not
        # build against NumPy 2)
        import contextlib
        with (ae.config.change_flags(cxx="") if name in REF_PY else contextlib.nullcontext()):
            f_ref = ae.function(ins, outs, mode=Mode("py", "fast_run") if name in REF_PY else REF_MODE,
                                on_unused_input="ignore", accept_inplace=True)
            ref_out = [np.asarray(o) for o in f_ref(*xs)]
        # 3. HIP lowering + oracle
        linker = HipLinker(executor_factory=lambda plan: (lambda *a: interp.run_plan(plan, a)))
        f_hip = ae.function(ins, outs, mode=Mode(linker, HIP_QUERY), on_unused_input="ignore",
                            accept_inplace=True)
        plan = f_hip.maker.linker.plan
        plan.name = name
        ora = [np.asarray(o) for o in f_hip(*xs)]
        is32 = any(o.dtype == np.float32 for o in ref_out)
        rt = rtol if rtol is not None else (1e-5 if is32 else 1e-12)
        atl = atol if atol is not None else (1e-6 if is32 else 1e-12)
        for k, (r, o) in enumerate(zip(ref_out, ora)):
            ok, msg = _close(o, r, exact, rt, atl)
            if not ok:
                raise SystemExit(f"ORACLE != REFERENCE for case {name} output {k}: {msg}\n"
                                 f"{plan.pretty()}")
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"),
                            **{f"out{k}": r for k, r in enumerate(ref_out)})
        index[name] = {"name": name, "ref_linker": "py" if name in REF_PY else "cvm", "plan": plan.to_json(), "inputs": specs, "exact": exact,
                       "rtol": rt, "atol": atl, "n_out": len(ref_out)}
        n_ok += 1
        print(f"[ok] {name}: {len(plan.nodes)} nodes, {len(ref_out)} outputs")
    with open(cases_path, "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "reference": "aesara-devs/aesara@2024-10-08",
                   "ref_mode": "Mode('cvm','fast_run')", "cases": list(index.values())}, f)
    print(f"{n_ok} cases written to {GOLDEN}")


if __name__ == "__main__":
    main()
