"""ORACLE — CPU (NumPy) restatement of the reference's per-Op algorithms for the hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package ``aesara_amd`` never does
(its device path fails loudly when the HIP library is missing).

Parity pinning: the reference holds no golden vectors for this path (every
numeric test compares against a NumPy restatement with tolerances).  This oracle is therefore
pinned against *outputs of the reference itself run in the authoring container* —
``oracle/gen_golden.py`` compiles each fixture graph with the reference's own C/py linkers
(``Mode("cvm", "fast_run")``) and stores the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this interpreter against those vectors (bit-exact for
integer/index ops, reference tolerances tensor/math.py:83-96 for floating point).

``run_plan(plan, inputs)`` interprets a :class:`aesara_amd.plan.Plan` (plain data; no Aesara).
"""
from __future__ import annotations

import math

import numpy as np

# ----------------------------------------------------------------------------------------
# scalar ops — reference: aesara/scalar/basic.py (`impl` / `c_code` of each ScalarOp)
# ----------------------------------------------------------------------------------------


def _softplus(x):
    # scalar/math.py:1133 Softplus.c_code thresholds (-37, 18, 33.3)
    x = np.asarray(x)
    with np.errstate(over="ignore", under="ignore"):
        return np.where(x < -37.0, np.exp(x),
                        np.where(x < 18.0, np.log1p(np.exp(np.minimum(x, 18.0))),
                                 np.where(x < 33.3, x + np.exp(-x), x))).astype(x.dtype)


def _ultra_fast_sigmoid(x):
    # tensor/nnet/sigm.py:54 UltraFastScalarSigmoid.c_code: x and z are variables of the output
    # dtype, the expressions between them are evaluated in double
    x = np.asarray(x)
    dt = x.dtype
    hx = (0.5 * x.astype(np.float64)).astype(dt)
    a = np.abs(hx).astype(np.float64)
    z = np.where(a < 1.7, 1.5 * a / (1 + a),
                 np.where(a < 3, 0.935409070603099 + 0.0458812946797165 * (a - 1.7), 0.99505475368673))
    z = np.where(hx >= 0, z, -z).astype(dt)
    return (0.5 * (z.astype(np.float64) + 1.0)).astype(dt)


def _sigmoid(x):
    # scalar/math.py:1110 Sigmoid.c_code: 1/(1+exp(-x))
    x = np.asarray(x)
    with np.errstate(over="ignore"):
        return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def _log1mexp(x):
    # scalar/math.py Log1mexp: x < -log(2) ? log1p(-exp(x)) : log(-expm1(x))
    x = np.asarray(x)
    with np.errstate(all="ignore"):
        return np.where(x < -0.6931471805599453, np.log1p(-np.exp(x)), np.log(-np.expm1(x)))


def _erf(x):
    from scipy import special
    return special.erf(x)


def _erfc(x):
    from scipy import special
    return special.erfc(x)


def _psi_as103(x):
    # scalar/math.py:361 Psi.c_support_code (AS 103); what the reference's C linker computes
    x = np.asarray(x)
    out = np.zeros(x.shape, "float64")
    for idx in np.ndindex(x.shape):
        y = float(x[idx])
        psi = 0.0
        if y <= 0.0:
            out[idx] = 0.0
            continue
        if y <= 1.0e-5:
            out[idx] = -0.5772156649 - 1.0 / y
            continue
        while y < 8.5:
            psi = psi - 1.0 / y
            y = y + 1
        R = 1.0 / y
        psi = psi + np.log(y) - .5 * R
        R = R * R
        psi = psi - R * (8.333333333e-2 - R * (8.333333333e-3 - R * 3.968253968e-3))
        out[idx] = psi
    return out.astype(x.dtype if x.dtype.kind == "f" else "float64")


def _trigamma_as121(x):
    # scalar/math.py:454 TriGamma.c_support_code (AS 121)
    x = np.asarray(x)
    out = np.zeros(x.shape, "float64")
    for idx in np.ndindex(x.shape):
        v = float(x[idx])
        if v <= 0:
            continue
        if v <= 0.0001:
            out[idx] = 1.0 / v / v
            continue
        value, z = 0.0, v
        while z < 5.0:
            value += 1.0 / z / z
            z += 1.0
        y = 1.0 / z / z
        value += 0.5 * y + (1.0 + y * (0.1666666667 + y * (-0.03333333333 + y * (0.02380952381 + y * -0.03333333333)))) / z
        out[idx] = value
    return out.astype(x.dtype if x.dtype.kind == "f" else "float64")


def _igam(k, x, upper):
    """scalar/c_code/gamma.c upperGamma / lowerGamma (what GammaU / GammaL's C bodies call): the
    continued fraction (modified Lentz, :_cfrac) / the power series (:_series), whatever x is, times
    exp(k log x - x); NaN for k <= 0 or x <= 0.  Element by element like the C loop."""
    k, x = np.broadcast_arrays(np.asarray(k, "float64"), np.asarray(x, "float64"))
    out = np.empty(k.shape, "float64")
    eps = 2.2204460492503131e-16
    tiny = eps * eps * eps
    one = np.float64(1.0)                         # NumPy scalars: 1 / 0 is inf like in C, no exception
    for idx in np.ndindex(k.shape):
        n, xx = np.float64(k[idx]), np.float64(x[idx])
        if not (n > 0) or not (xx > 0):
            out[idx] = np.nan
            continue
        if upper:
            b = xx + 1 - n
            c = one / tiny
            d = one / b
            f = d
            for i in range(1, 1024):
                a = i * (n - i)
                b += 2
                d = a * d + b
                if abs(d) < tiny:
                    d = tiny
                c = b + a / c
                if abs(c) < tiny:
                    c = tiny
                d = one / d
                e = d * c
                f *= e
                if abs(e - 1) < eps:
                    break
            val = f
        else:
            t = s_ = one / n
            m = n
            for _ in range(1024):
                m += 1
                t *= xx / m
                s_ += t
                if abs(t) < abs(s_) * eps:
                    break
            val = s_
        out[idx] = val * np.exp(n * np.log(xx) - xx)
    return out


def _sp():
    from scipy import special
    return special


def _special(name):
    def f(x):
        from scipy import special
        return getattr(special, name)(x)
    return f


def _round_away(x):
    # scalar/basic.py:2799 RoundHalfAwayFromZero
    return np.where(x < 0, np.ceil(x - 0.5), np.floor(x + 0.5))


_UNARY = {
    "neg": np.negative, "abs": np.abs, "sgn": np.sign, "sqr": np.square, "sqrt": np.sqrt,
    "exp": np.exp, "exp2": np.exp2, "expm1": np.expm1, "log": np.log, "log2": np.log2,
    "log10": np.log10, "log1p": np.log1p, "sin": np.sin, "cos": np.cos, "tan": np.tan,
    "arcsin": np.arcsin, "arccos": np.arccos, "arctan": np.arctan, "sinh": np.sinh,
    "cosh": np.cosh, "tanh": np.tanh, "arcsinh": np.arcsinh, "arccosh": np.arccosh,
    "arctanh": np.arctanh, "ceil": np.ceil, "floor": np.floor, "trunc": np.trunc,
    "round_half_to_even": np.around, "round_half_away_from_zero": _round_away,
    "reciprocal": np.reciprocal, "identity": lambda x: x, "invert": np.invert,
    "isnan": np.isnan, "isinf": np.isinf, "sigmoid": _sigmoid, "softplus": _softplus,
    "erf": _erf, "erfc": _erfc, "log1mexp": _log1mexp, "deg2rad": np.deg2rad,
    "rad2deg": np.rad2deg,
    "erfcx": _special("erfcx"), "erfinv": _special("erfinv"), "erfcinv": _special("erfcinv"),
    "gamma": _special("gamma"), "gammaln": _special("gammaln"), "psi": _psi_as103,
    "tri_gamma": _trigamma_as121, "j0": _special("j0"), "j1": _special("j1"),
    "i0": _special("i0"), "i1": _special("i1"),
    "softsign": lambda x: x / (1.0 + np.abs(x)),     # tensor/nnet/basic.py:2048
    "xlogx": lambda x: np.where(x == 0, 0, x * np.log(x)),       # tensor/xlogx.py:15 XlogX.impl
    "ultra_fast_sigmoid": _ultra_fast_sigmoid,
}

_BINARY = {
    "sub": np.subtract, "true_div": np.true_divide, "int_div": np.floor_divide,
    "mod": np.mod, "pow": np.power, "arctan2": np.arctan2,
    "xlogy0": lambda x, y: np.where(x == 0, 0, x * np.log(y)),   # tensor/xlogx.py:44 XlogY0.impl
    # scalar/math.py:580 GammaInc.st_impl ... :877 GammaL.st_impl (SciPy, like the reference's impl)
    "gammainc": lambda k, x: _sp().gammainc(k, x), "gammaincc": lambda k, x: _sp().gammaincc(k, x),
    "chi2sf": lambda x, k: _sp().gammaincc(k / 2.0, x / 2.0),
    "gammau": lambda k, x: _igam(k, x, True), "gammal": lambda k, x: _igam(k, x, False),
    "lt": np.less, "gt": np.greater, "le": np.less_equal, "ge": np.greater_equal,
    "eq": np.equal, "neq": np.not_equal,
}

_NARY = {"add": np.add, "mul": np.multiply, "maximum": np.maximum, "minimum": np.minimum,
         "and": np.bitwise_and, "or": np.bitwise_or, "xor": np.bitwise_xor}

_FLOAT_FUNCS = {"sqrt", "exp", "exp2", "expm1", "log", "log2", "log10", "log1p", "sin", "cos",
                "tan", "arcsin", "arccos", "arctan", "sinh", "cosh", "tanh", "arcsinh",
                "arccosh", "arctanh", "sigmoid", "softplus", "erf", "erfc", "log1mexp",
                "deg2rad", "rad2deg", "reciprocal", "true_div", "arctan2", "erfcx", "erfinv",
                "erfcinv", "gamma", "gammaln", "psi", "tri_gamma", "j0", "j1", "i0", "i1", "softsign",
                "xlogx", "xlogy0", "gammainc", "gammaincc", "chi2sf", "gammau", "gammal", "ultra_fast_sigmoid"}


def eval_scalar_expr(s, ins):
    """Evaluate a plan scalar expression on broadcast-compatible ndarrays ``ins``.

    Mirrors ``Composite.py_perform`` (scalar/basic.py:4075) / the per-temp C code of
    ``c_code_template`` (:4250): one temporary per scalar node, each stored in the node's
    declared dtype.
    """
    temps = []

    def get(r):
        if r[0] == "i":
            return ins[r[1]]
        if r[0] == "t":
            return temps[r[1]]
        return np.asarray(r[1], dtype=r[2])

    with np.errstate(all="ignore"):
        for n in s["nodes"]:
            op, dt = n["op"], np.dtype(n["dtype"])
            a = [get(r) for r in n["in"]]
            if op in _FLOAT_FUNCS and dt.kind == "f":
                a = [x.astype(dt) if x.dtype != dt and op not in ("true_div", "arctan2") else x
                     for x in a]
            if op in _UNARY:
                r = _UNARY[op](a[0])
            elif op in _BINARY:
                if op in ("true_div", "arctan2"):
                    r = _BINARY[op](a[0].astype(dt), a[1].astype(dt))
                elif op in ("int_div", "mod") and dt.kind in "iu":
                    # integer x // 0 and x % 0: reference C raises ZeroDivisionError
                    # (scalar/basic.py:2068, :2186); NumPy/our device path return 0.
                    r = _BINARY[op](a[0], a[1])
                else:
                    r = _BINARY[op](a[0], a[1])
            elif op in _NARY:
                r = a[0]
                if op in ("add", "mul"):
                    r = r.astype(dt)
                for x in a[1:]:
                    r = _NARY[op](r, x.astype(dt) if op in ("add", "mul") else x)
            elif op == "cast":
                r = a[0].astype(dt)
            elif op == "second":
                r = np.broadcast_arrays(a[0], a[1])[1]
            elif op == "switch":
                r = np.where(a[0] != 0, a[1], a[2])
            elif op == "clip":
                # scalar/basic.py:2342 Clip: x < min ? min : x > max ? max : x
                r = np.where(a[0] < a[1], a[1], np.where(a[0] > a[2], a[2], a[0]))
            else:
                raise NotImplementedError(f"oracle: scalar op {op}")
            temps.append(np.asarray(r).astype(dt, copy=False))
    return [get(r) for r in s["out"]]


# ----------------------------------------------------------------------------------------
# tensor ops
# ----------------------------------------------------------------------------------------
_REDUCE = {"add": np.add, "mul": np.multiply, "maximum": np.maximum, "minimum": np.minimum,
           "and": np.bitwise_and, "or": np.bitwise_or, "xor": np.bitwise_xor}


def careduce(x, scalar_op, axis, acc_dtype, out_dtype):
    """reference: tensor/elemwise.py:1495 CAReduce.perform — ``ufunc.reduce`` applied axis by
    axis with ``dtype=acc_dtype`` and a final cast to the output dtype (:1506-1513)."""
    if axis is None:
        axis = list(range(x.ndim))
    if scalar_op == "mul_without_zeros":
        # tensor/math.py:2713 MulWithoutZeros (identity 0, zeros skipped): the product of the
        # non-zero entries, 0 when there is none
        xa = np.asarray(x).astype(acc_dtype)
        ax = tuple(axis)
        v = np.where(xa == 0, 1, xa).prod(axis=ax) * (xa != 0).any(axis=ax)
        return np.asarray(v, dtype=acc_dtype).astype(out_dtype)
    ufunc = _REDUCE[scalar_op]
    v = x
    acc = np.dtype(acc_dtype)
    for a in sorted(axis, reverse=True):
        if v.shape[a] == 0 and scalar_op in ("maximum", "minimum"):
            raise ValueError("zero-size array to reduction operation which has no identity")
        if acc.kind == "b" or scalar_op in ("maximum", "minimum"):
            v = ufunc.reduce(v, a)
        else:
            v = ufunc.reduce(v, a, dtype=acc)
    return np.asarray(v, dtype=acc).astype(out_dtype)


def dimshuffle(x, new_order):
    """reference: tensor/elemwise.py:222 DimShuffle.perform (transpose, drop, then expand)."""
    keep = [d for d in new_order if d != "x"]
    dropped = [d for d in range(x.ndim) if d not in keep]
    for d in dropped:
        if x.shape[d] != 1:
            raise ValueError("Cannot drop a non-broadcastable dimension")
    v = x.transpose(keep + dropped)
    shape = [v.shape[keep.index(d)] if d != "x" else 1 for d in new_order]
    return v.reshape(shape)


def gemm(z, a, x, y, b):
    """reference: tensor/blas.py:984 Gemm.perform — z*b + a*dot(x, y) (b==0 ignores z)."""
    dt = z.dtype
    a = np.asarray(a, dt)
    b = np.asarray(b, dt)
    if b == 0.0:
        r = a * np.dot(x, y) if a != 1.0 else np.dot(x, y)
    else:
        r = b * z + a * np.dot(x, y)
    if np.ndim(z) == 2:          # :995 z broadcast up to the product, the product up to z
        r = np.broadcast_to(r, (max(z.shape[0], x.shape[0]), max(z.shape[1], y.shape[1])))
    return np.array(r, dtype=dt)


def gemv(y, alpha, A, x, beta):
    """reference: tensor/blas.py:279 Gemv.perform — beta*y + alpha*dot(A, x)."""
    dt = y.dtype
    out = np.asarray(alpha, dt) * np.dot(A, x)
    if np.asarray(beta) != 0:
        out = out + np.asarray(beta, dt) * y
    return np.asarray(out, dtype=dt)


def _resolve_idx(idx_list, extra):
    """reference: tensor/subtensor.py:756 Subtensor.perform / get_idx_list."""
    extra = list(extra)

    def ent(e):
        if e == "in":
            return int(np.asarray(extra.pop(0)))
        return e

    out = []
    for e in idx_list:
        if "slice" in e:
            st, sp, se = (ent(t) for t in e["slice"])
            out.append(slice(st, sp, se))
        else:
            out.append(ent(e["index"]))
    assert not extra
    return tuple(out)


def scan(p, inputs, run_inner):
    """reference: scan/op.py:1673 Scan.perform / scan_perform.pyx:71 — sequences, mit-mot
    (op.py:1954-1991), mit-sot / sit-sot with circular buffers of length ``store_steps``, nit-sot,
    shared outputs (op.py:1833-1844, 2094-2100), non-sequences and the do-while condition
    (op.py:1947-1949, truncation :2139-2159)."""
    n_seqs = p["n_seqs"]
    mm_in = [list(t) for t in p.get("mit_mot_in_slices", [])]
    mm_out = [list(t) for t in p.get("mit_mot_out_slices", [])]
    mit = p["mit_sot_in_slices"]
    sit = p["sit_sot_in_slices"]
    n_nit = p["n_nit_sot"]
    n_sh = p.get("n_shared_outs", 0)
    as_while = p.get("as_while", False)
    n_steps = int(np.asarray(inputs[0]))
    if n_steps < 0:
        raise IndexError(f"Scan was asked to run for negative number of step {n_steps}")
    seqs = inputs[1:1 + n_seqs]
    for k, sq in enumerate(seqs):
        if sq.shape[0] < n_steps:
            raise ValueError(f"Sequence {k} has shape {sq.shape} but the Scan's required "
                             f"number of steps is {n_steps}")
    n_mm = len(mm_in)
    taps = mm_in + [list(t) for t in mit] + [list(t) for t in sit]
    n_rec = len(taps)
    o = 1 + n_seqs
    rec_init = inputs[o:o + n_rec]
    shared = list(inputs[o + n_rec:o + n_rec + n_sh])
    o += n_rec + n_sh
    nit_len = [int(np.asarray(v)) for v in inputs[o:o + n_nit]]
    non_seqs = inputs[o + n_nit:]
    mintaps = [min(t) for t in taps] + [0] * n_nit
    outs = [np.array(v, copy=True) for v in rec_init] + [None] * n_nit
    store = [v.shape[0] for v in rec_init] + nit_len
    if n_steps == 0:
        # op.py:1753-1762: nit-sot outputs are empty, shared outputs are left unset (None)
        return outs[:n_rec] + [None] * n_nit + [None] * n_sh
    pos = [(-mintaps[k]) % store[k] for k in range(n_rec + n_nit)]
    n_mm_outs = sum(len(t) for t in mm_out)
    i, cond = 0, True
    while i < n_steps and cond:
        args = [sq[i] for sq in seqs]
        for k in range(n_rec):
            for t in taps[k]:
                args.append(outs[k][(pos[k] + t) % store[k]].copy())
        args.extend(shared)
        args.extend(non_seqs)
        res = run_inner(args)
        if as_while:
            cond = np.asarray(res[n_mm_outs + (n_rec - n_mm) + n_nit + n_sh]) == 0
        ro = 0
        for g in range(n_mm):
            for sl in mm_out[g]:
                outs[g][sl + pos[g]] = res[ro]
                ro += 1
        for k in range(n_mm, n_rec):
            outs[k][pos[k]] = res[ro]
            ro += 1
        for j in range(n_nit):
            k = n_rec + j
            if i == 0:
                outs[k] = np.empty((store[k],) + np.shape(res[ro]), dtype=np.asarray(res[ro]).dtype)
            outs[k][pos[k]] = res[ro]
            ro += 1
        shared = [np.asarray(r) for r in res[ro:ro + n_sh]]
        pos = [(pp + 1) % st for pp, st in zip(pos, store)]
        i += 1
    # rotate circular buffers into chronological order (op.py:2105-2134); zero / truncate the
    # part a do-while that stopped early (or truncated BPTT) never wrote (op.py:2139-2159)
    for k in range(n_mm, n_rec + n_nit):
        if store[k] < i - mintaps[k] and pos[k] < store[k]:
            outs[k] = np.concatenate([outs[k][pos[k]:], outs[k][:pos[k]]], axis=0)
        elif store[k] > i - mintaps[k]:
            outs[k][i - mintaps[k]:] = 0
            if i < n_steps:
                outs[k] = outs[k][:-(n_steps - i)]
    return outs + shared


def argmax(x, axes):
    """reference: tensor/math.py:388 Argmax.perform — kept axes in front, reduced axes flattened
    last, np.argmax over the last axis, int64 result."""
    x = np.asarray(x)
    keep = [i for i in range(x.ndim) if i not in axes]
    t = np.transpose(x, keep + list(axes))
    kept_shape = t.shape[:len(keep)]
    flat = t.reshape(kept_shape + (int(np.prod(t.shape[len(keep):])),))
    return np.asarray(np.argmax(flat, axis=-1), dtype="int64")


def specify_shape_check(xshape, p, given):
    """reference: tensor/shape.py:439-450; ``p["dims"]`` = positions whose size is given
    (older plans carry no "dims": every given value is positional from 0)."""
    dims = p.get("dims", list(range(len(given))))
    ndim = p.get("ndim", len(xshape))
    want = [None] * ndim
    for d, s in zip(dims, given):
        want[d] = s
    if len(xshape) != ndim:
        raise AssertionError(f"SpecifyShape: Got {len(xshape)} dimensions (shape {tuple(xshape)}), "
                             f"expected {ndim} dimensions with shape {tuple(want)}.")
    if not all(xs == s for xs, s in zip(xshape, want) if s is not None):
        raise AssertionError(f"SpecifyShape: Got shape {tuple(xshape)}, expected {tuple(want)}.")


def adv_index(params, extra):
    """The NumPy index tuple of an AdvancedSubtensor / AdvancedIncSubtensor node: integer arrays
    only (no "index" param), or arrays mixed with slices / newaxis (tensor/subtensor.py:2543)."""
    if "index" not in params:
        return tuple(np.asarray(i) for i in extra)
    idx = []
    for e in params["index"]:
        if "array" in e:
            idx.append(np.asarray(extra[e["array"]]))
        elif "mask" in e:
            idx.append(np.asarray(extra[e["mask"]], dtype=bool))
        elif "slice" in e:
            idx.append(slice(*[c if (c is None or isinstance(c, int)) else int(np.asarray(extra[c["in"]]))
                               for c in e["slice"]]))
        else:
            idx.append(None)
    return tuple(idx)


def run_plan(plan, inputs):
    """Interpret ``plan`` on NumPy arrays; returns the list of outputs."""
    env = {}
    for vid, v in plan.vars.items():
        if v.const is not None:
            env[vid] = v.const_value()
    assert len(inputs) == len(plan.inputs), (len(inputs), len(plan.inputs))
    for vid, x in zip(plan.inputs, inputs):
        env[vid] = np.asarray(x)
    for n in plan.nodes:
        a = [env[i] for i in n.inputs]
        p = n.params
        op = n.op
        ov = [plan.vars[o] for o in n.outputs]
        if op == "Elemwise":
            # reference: tensor/elemwise.py:725 Elemwise.perform (shape check :733-735)
            nd = max((x.ndim for x in a), default=0)
            for d in range(nd):
                sizes = {x.shape[d] for x in a if x.shape[d] != 1}
                if len(sizes) > 1:
                    raise ValueError(f"Shapes on dimension {d} do not match")
            res = eval_scalar_expr(p["scalar"], a)
            shape = np.broadcast_shapes(*[x.shape for x in a]) if a else ()
            r = [np.broadcast_to(x, shape).astype(o.dtype) for x, o in zip(res, ov)]
        elif op == "CAReduce":
            r = [careduce(a[0], p["scalar_op"], p["axis"], p["acc_dtype"], ov[0].dtype)]
        elif op == "DimShuffle":
            r = [dimshuffle(a[0], p["new_order"])]
        elif op in ("Dot", "Dot22"):
            r = [np.asarray(np.dot(a[0], a[1]), dtype=ov[0].dtype)]
        elif op == "Dot22Scalar":
            r = [np.asarray(np.dot(a[0], a[1]) * a[2], dtype=ov[0].dtype)]
        elif op == "Gemm":
            r = [gemm(*a)]
        elif op == "Gemv":
            r = [gemv(*a)]
        elif op == "Ger":
            # reference: tensor/blas.py:330 Ger.perform: A + alpha * outer(x, y)
            r = [np.asarray(a[0] + a[1] * np.outer(a[2], a[3]), dtype=ov[0].dtype)]
        elif op == "BatchedDot":
            # reference: tensor/blas.py:2224 BatchedDot.perform
            # (z[i] = np.dot(x[i], y[i]); a 2-d operand is a batch of vectors, :2196-2207)
            x_, y_ = np.asarray(a[0]), np.asarray(a[1])
            if x_.shape[0] != y_.shape[0]:
                raise TypeError(f"Shape mismatch: x has {x_.shape[0]} rows but y has {y_.shape[0]} rows")
            xs_ = x_[:, None, :] if x_.ndim == 2 else x_
            ys_ = y_[:, :, None] if y_.ndim == 2 else y_
            z_ = np.matmul(xs_, ys_)
            if y_.ndim == 2:
                z_ = z_[:, :, 0]
            if x_.ndim == 2:
                z_ = z_[:, 0]
            r = [np.asarray(z_, dtype=ov[0].dtype)]
        elif op == "Alloc":
            # reference: tensor/basic.py:1427 Alloc.perform
            shape = tuple(int(np.asarray(s)) for s in a[1:])
            r = [np.array(np.broadcast_to(a[0], shape), dtype=ov[0].dtype)]
        elif op == "BroadcastTo":
            # reference: tensor/extra_ops.py BroadcastTo.perform (np.broadcast_to view)
            r = [np.broadcast_to(a[0], tuple(int(np.asarray(s)) for s in a[1:]))]
        elif op == "AllocEmpty":
            shape = tuple(int(np.asarray(s)) for s in a)
            r = [np.zeros(shape, dtype=p["dtype"])]  # contents unspecified in the reference
        elif op == "MakeVector":
            r = [np.array([np.asarray(x) for x in a], dtype=p["dtype"])]
        elif op == "Join":
            ax = int(np.asarray(a[0]))
            r = [np.concatenate(a[1:], axis=ax).astype(ov[0].dtype)]
        elif op == "SpecifyShape":
            # reference: tensor/shape.py:439 SpecifyShape.perform
            specify_shape_check(np.shape(a[0]), p, [int(np.asarray(v)) for v in a[1:]])
            r = [a[0]]
        elif op in ("ScalarFromTensor", "TensorFromScalar", "ViewOp"):
            r = [a[0]]
        elif op == "IfElse":
            # reference: ifelse.py:61 IfElse (the oracle evaluates both branches, then selects)
            no = p["n_outs"]
            r = list(a[1:1 + no]) if bool(np.asarray(a[0])) else list(a[1 + no:1 + 2 * no])
        elif op == "Assert":
            # reference: raise_op.py:94 CheckAndRaise.perform
            if not np.all([np.asarray(c) for c in a[1:]]):
                import builtins
                exc = getattr(builtins, p.get("exc_type", "AssertionError"), AssertionError)
                raise exc(p.get("msg", ""))
            r = [a[0]]
        elif op == "Split":
            # reference: tensor/basic.py:1929 Split.perform
            x, axis, splits = a[0], int(np.asarray(a[1])), [int(v) for v in np.asarray(a[2])]
            if len(splits) != p["len_splits"]:
                raise ValueError("Length of `splits` is not equal to `len_splits`")
            if sum(splits) != x.shape[axis]:
                raise ValueError(f"The splits sum to {sum(splits)}; expected {x.shape[axis]}")
            if any(nb < 0 for nb in splits):
                raise ValueError("Attempted to make an array with a negative number of elements")
            r, lo = [], 0
            key = [slice(None)] * x.ndim
            for nb in splits:
                key[axis] = slice(lo, lo + nb)
                r.append(x[tuple(key)].copy())
                lo += nb
        elif op == "CumOp":
            # reference: tensor/extra_ops.py:311 CumOp.perform (result in the output dtype)
            fn = np.cumsum if p["mode"] == "add" else np.cumprod
            r = [fn(a[0], axis=p["axis"], dtype=ov[0].dtype)]
        elif op in ("Sort", "ArgSort"):
            # reference: tensor/sort.py:48 SortOp.perform / :184 ArgSortOp.perform.  Ties in input
            # order (kind="stable"): the only defined order; equal to the default introsort's result
            # whenever the keys are distinct
            ax = None if a[1] is None or np.asarray(a[1]).dtype == object else int(np.asarray(a[1]))
            fn = np.sort if op == "Sort" else np.argsort
            r = [np.asarray(fn(np.asarray(a[0]), axis=ax, kind="stable"), dtype=ov[0].dtype)]
        elif op == "Default":
            # reference: tensor/basic.py:1819 Default.perform
            r = [np.array(a[1], copy=True) if a[0] is None or (a[0].dtype == object and a[0].ndim == 0
                                                             and a[0].item() is None) else a[0]]
        elif op == "Searchsorted":
            # reference: tensor/extra_ops.py:144 SearchsortedOp.perform
            r = [np.searchsorted(a[0], a[1], side=p["side"],
                                 sorter=a[2] if len(a) > 2 else None).astype("int64")]
        elif op == "HostCall":
            # reference: printing.py:863 Print.perform / compile/ops.py:258 FromFunctionOp.perform
            res = p["fn"](*a)
            r = [a[0]] if p.get("view") else [np.asarray(x) for x in res]
        elif op == "LexArgSortRows":
            # the order np.unique(x, axis=k) gives its items (tensor/extra_ops.py:1216): rows as
            # records, first column most significant; np.lexsort is stable, last key primary
            xx = np.asarray(a[0])
            r = [np.lexsort(xx.T[::-1]).astype("int64") if xx.shape[1] else np.arange(xx.shape[0], dtype="int64")]
        elif op == "Nonzero":
            # reference: tensor/basic.py:870 Nonzero.perform
            r = [np.asarray(i, dtype="int64") for i in np.nonzero(np.asarray(a[0]))]
        elif op == "FillDiagonal":
            # reference: tensor/extra_ops.py:906 FillDiagonal.perform
            aa = np.array(a[0], copy=True)
            if aa.ndim == 2:
                aa.flat[:aa.shape[1] * aa.shape[1]:aa.shape[1] + 1] = a[1]
            else:
                np.fill_diagonal(aa, a[1])
            r = [aa]
        elif op == "MatMul":
            # reference: tensor/math.py:2941 MatMul.perform
            r = [np.matmul(a[0], a[1])]
        elif op == "Eye":
            # reference: tensor/basic.py:1278 Eye.perform
            r = [np.eye(int(np.asarray(a[0])), int(np.asarray(a[1])), int(np.asarray(a[2])), dtype=p["dtype"])]
        elif op == "Tri":
            # reference: tensor/basic.py:1000 Tri.perform
            r = [np.tri(int(np.asarray(a[0])), int(np.asarray(a[1])), int(np.asarray(a[2])), dtype=p["dtype"])]
        elif op == "ExtractDiag":
            # reference: tensor/basic.py:3402 ExtractDiag.perform
            r = [np.array(np.asarray(a[0]).diagonal(p["offset"], p["axis1"], p["axis2"]), copy=True)]
        elif op == "AllocDiag":
            # reference: tensor/basic.py:3523 AllocDiag.perform
            xx = np.asarray(a[0])
            ax1, ax2 = min(p["axis1"], p["axis2"]), max(p["axis1"], p["axis2"])
            off = p["offset"]
            res = np.zeros(xx.shape[:-1] + (xx.shape[-1] + abs(off),) * 2, dtype=xx.dtype)
            ids = np.arange(xx.shape[-1])
            res[(Ellipsis, ids + max(0, -off), ids + max(0, off))] = xx
            if xx.ndim > 1:
                axes = list(range(xx.ndim - 1))
                last = axes[-1]
                axes = axes[:ax1] + [last + 1] + axes[ax1:]
                axes = axes[:ax2] + [last + 2] + axes[ax2:]
                res = res.transpose(axes)
            r = [res]
        elif op == "ARange":
            # reference: tensor/basic.py:2937 ARange.perform
            st, sp, se = (np.asarray(v).item() for v in a)
            r = [np.arange(st, sp, se, dtype=p["dtype"])]
        elif op == "AdvancedSubtensor":
            # reference: tensor/subtensor.py:2607 AdvancedSubtensor.perform
            r = [np.asarray(a[0])[adv_index(p, a[1:])]]
        elif op == "AdvancedIncSubtensor":
            # reference: tensor/subtensor.py:2688 AdvancedIncSubtensor.perform
            out = np.array(a[0], copy=True)
            idx = adv_index(p, a[2:])
            if p["set_instead_of_inc"]:
                out[idx] = a[1]
            elif p.get("ignore_duplicates"):
                out[idx] += a[1]              # :2693 (buffered: read, add, sequential set)
            else:
                np.add.at(out, idx, a[1])
            r = [out]
        elif op == "Argmax":
            r = [argmax(a[0], p["axis"])]
        elif op == "DeepCopyOp":
            r = [np.array(a[0], copy=True)]
        elif op == "Shape_i":
            r = [np.asarray(a[0].shape[p["i"]], dtype="int64")]
        elif op == "Shape":
            r = [np.asarray(a[0].shape, dtype="int64")]
        elif op == "Reshape":
            shp = tuple(int(s) for s in np.asarray(a[1]).reshape(-1))
            if p.get("ndim") is not None and len(shp) != p["ndim"]:     # tensor/shape.py:649
                raise ValueError("Shape argument to Reshape has incorrect length: "
                                 f"{len(shp)}, should be {p['ndim']}")
            r = [np.reshape(a[0], shp)]
        elif op == "Subtensor":
            r = [a[0][_resolve_idx(p["idx_list"], a[1:])]]
        elif op == "IncSubtensor":
            # reference: tensor/subtensor.py:1556 IncSubtensor.perform
            x = np.array(a[0], copy=True)
            idx = _resolve_idx(p["idx_list"], a[2:])
            if p["set_instead_of_inc"]:
                x[idx] = a[1]
            else:
                x[idx] += a[1]
            r = [x]
        elif op == "AdvancedSubtensor1":
            # reference: tensor/subtensor.py:1953 (x.take(i, axis=0)); IndexError when out of range
            r = [a[0].take(a[1], axis=0)]
        elif op == "AdvancedIncSubtensor1":
            # reference: tensor/subtensor.py:2128 (np.add.at semantics for inc)
            x = np.array(a[0], copy=True)
            if p["set_instead_of_inc"]:
                x[a[2]] = a[1]
            else:
                np.add.at(x, a[2], a[1])
            r = [x]
        elif op == "Scan":
            inner = p["inner"]
            r = scan(p, a, lambda args: run_plan(inner, args))
        else:
            raise NotImplementedError(f"oracle: op {op}")
        for o, val in zip(n.outputs, r):
            env[o] = val
    return [env[o] for o in plan.outputs]
