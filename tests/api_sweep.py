"""API sweep (authoring container only; needs /root/reference): typical tensor-API expressions
compiled through the HIP linker's lowering + the NumPy oracle as executor, compared with the
reference's own C linker.  Not collected by pytest (compiling ~150 reference functions takes
minutes); run `python tests/api_sweep.py` — the report of the last run is
profiles/r01_api_sweep.txt.  TEST INFRASTRUCTURE: this is how gaps in the lowering are found."""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import ref_overlay  # noqa: E402

ae = ref_overlay.import_reference()
import numpy as np  # noqa: E402
import aesara.tensor as at  # noqa: E402
from aesara.compile.mode import Mode  # noqa: E402

import interp  # noqa: E402
from aesara_amd.linker import HIP_QUERY, HipLinker  # noqa: E402

logging.getLogger("aesara").setLevel(logging.CRITICAL)


def sweep1():
    """tensor API"""
    x = at.dmatrix("x"); y = at.dmatrix("y"); v = at.dvector("v"); t3 = at.dtensor3("t3")
    i = at.lvector("i"); s = at.dscalar("s"); n = at.lscalar("n")
    xv = np.random.default_rng(0).standard_normal((5, 4)); yv = np.random.default_rng(1).standard_normal((5, 4))
    vv = np.random.default_rng(2).standard_normal(4); tv = np.random.default_rng(3).standard_normal((3, 5, 4))
    iv = np.array([0, 2, 1, 3]); sv = 1.7; nv = 3
    VALS1 = dict(zip([x, y, v, t3, i, s, n], [xv, yv, vv, tv, iv, sv, nv]))
    E1 = {
     "mean": lambda: x.mean(axis=0), "var": lambda: x.var(axis=1), "std": lambda: x.std(), "logsumexp": lambda: at.logsumexp(x, axis=1),
     "clip": lambda: at.clip(x, -0.5, 0.5), "where": lambda: at.where(x > 0, x, y), "tile": lambda: at.tile(x, (2, 3)), "repeat": lambda: at.repeat(x, 2, axis=0),
     "eye": lambda: at.eye(n, n) * s, "tri": lambda: at.tril(x) + at.triu(y), "diag_extract": lambda: at.diag(x[:4]), "diag_alloc": lambda: at.diag(v),
     "concatenate": lambda: at.concatenate([x, y], axis=1), "stack": lambda: at.stack([x, y]), "flatten": lambda: x.flatten() * 2, "transpose": lambda: x.T + 1,
     "swapaxes": lambda: at.swapaxes(t3, 0, 2) * 2, "moveaxis": lambda: at.moveaxis(t3, 0, -1) + 0, "squeeze": lambda: at.squeeze(x[:1]) + 0, "expand_dims": lambda: at.expand_dims(v, 0) * x,
     "roll": lambda: at.roll(x, 1, axis=0), "sort": lambda: at.sort(v), "argsort": lambda: at.argsort(v), "isnan": lambda: at.isnan(x) | at.isinf(y),
     "sign_abs_round": lambda: at.sgn(x) * abs(y) + at.round(x) + at.floor(y) + at.ceil(x), "pow": lambda: abs(x) ** y + at.sqrt(abs(x)), "log1p_expm1": lambda: at.log1p(abs(x)) + at.expm1(y),
     "erf": lambda: at.erf(x) + at.erfc(y), "gammaln": lambda: at.gammaln(abs(x) + 1), "sigmoid_softplus": lambda: at.sigmoid(x) + at.softplus(y) + at.tanh(x), "arctan2": lambda: at.arctan2(x, y),
     "maxmin": lambda: at.maximum(x, y) - at.minimum(x, y), "prod_all_any": lambda: [x.prod(axis=0), at.all(x > 0, axis=1), at.any(x > 0)], "ptp": lambda: at.ptp(x, axis=0),
     "outer": lambda: at.outer(v, v), "tensordot": lambda: at.tensordot(t3, x, axes=[[1, 2], [0, 1]]), "batched_dot": lambda: at.batched_dot(t3, t3.dimshuffle(0, 2, 1)),
     "matmul": lambda: at.matmul(t3, x.T), "dot3d": lambda: at.dot(t3, v), "norm": lambda: x.norm(2),
     "cumsum": lambda: at.cumsum(x, axis=0), "cumprod": lambda: at.cumprod(x, axis=1), "diff": lambda: at.diff(x, axis=0), "linspace": lambda: at.linspace(0, 1, 7) * s,
     "arange": lambda: at.arange(n) * s, "zeros_like": lambda: at.zeros_like(x) + at.ones_like(y), "full": lambda: at.full((n, n), s), "alloc": lambda: at.alloc(v, 3, 4),
     "fill": lambda: at.fill(x, s), "one_hot": lambda: at.extra_ops.to_one_hot(i, 5), "argmin_min": lambda: [at.argmin(x, axis=0), at.min(x, axis=1)], "searchsorted": lambda: at.extra_ops.searchsorted(at.sort(v), v),
     "bincount": lambda: at.extra_ops.bincount(i), "take": lambda: at.take(x, i, axis=1), "take_along": lambda: at.take_along_axis(x, at.argsort(x, axis=1), axis=1) if hasattr(at, "take_along_axis") else x,
     "switch": lambda: at.switch(x > y, x, y), "isclose": lambda: at.isclose(x, y), "logaddexp": lambda: at.logaddexp(x, y), "log_softmax": lambda: at.special.log_softmax(x, axis=1),
     "softmax0": lambda: at.special.softmax(x, axis=0), "grad_mean_var": lambda: ae.grad((x.var(axis=0) + x.mean(axis=1).sum()).sum(), x), "grad_softmax": lambda: ae.grad((at.special.softmax(x, axis=1) * y).sum(), x),
     "grad_dot": lambda: ae.grad(at.dot(x, y.T).sum() + at.dot(x, v).sum(), [x, v]), "grad_subtensor": lambda: ae.grad((x[1:3, ::2] ** 2).sum() + x[i[:2] % 5].sum(), x), "grad_max": lambda: ae.grad(x.max(axis=0).sum(), x),
     "grad_norm": lambda: ae.grad(x.norm(2), x), "grad_concat": lambda: ae.grad((at.concatenate([x, y], axis=0) ** 2).sum(), [x, y]), "grad_cumsum": lambda: ae.grad((at.cumsum(x, axis=0) ** 2).sum(), x),
     "grad_clip_switch": lambda: ae.grad((at.clip(x, -0.5, 0.5) * at.switch(x > 0, y, -y)).sum(), [x, y]), "int_div_mod": lambda: (i // 2) + (i % 3) + (i ** 2), "cast": lambda: at.cast(x, "float32") * 2,
     "reshape": lambda: x.reshape((4, 5)) + 0, "reshape_sym": lambda: x.reshape((n + 1, -1)) * 1, "shape_ops": lambda: x.shape[0] * x.shape[1] + n, "mgrid_like": lambda: at.arange(5)[:, None] * at.arange(4)[None, :] + x,
     "inv_sqrt": lambda: 1 / at.sqrt(1 + x * x), "neq": lambda: at.neq(x, y) & at.le(x, 0), "bitwise": lambda: (i & 3) | (i ^ 5) | (~i), "floor_div_float": lambda: x // 0.3 + x % 0.3,
     "sum_keepdims": lambda: x / x.sum(axis=1, keepdims=True), "mean_keepdims": lambda: x - x.mean(axis=0, keepdims=True), "unique": lambda: at.extra_ops.unique(i),
     "triu_indices": lambda: x[at.triu(at.ones((5, 4))).nonzero()], "nonzero": lambda: at.nonzero(v > 0)[0], "set_subtensor_bool": lambda: at.set_subtensor(x[x > 0], 0.0),
     "broadcast_to": lambda: at.broadcast_to(v, (3, 4)) + 0, "kron": lambda: at.slinalg.kron(x[:2, :2], y[:2, :2]), "cholesky": lambda: at.slinalg.cholesky(at.dot(x.T, x) + at.eye(4)),
     "solve": lambda: at.slinalg.solve(at.dot(x.T, x) + at.eye(4), v), "matrix_inverse": lambda: at.nlinalg.matrix_inverse(at.dot(x.T, x) + at.eye(4)), "det": lambda: at.nlinalg.det(at.dot(x.T, x)),
     "random_normal": lambda: at.random.normal(0, 1, size=(3,)), "scan_cumsum": lambda: ae.scan(lambda r, a: a + r, sequences=[x], outputs_info=[at.zeros_like(x[0])])[0],
    }
    return E1, VALS1


def sweep2():
    """dtypes, gradients, scans"""
    rng = np.random.default_rng(0)
    x = at.dmatrix("x"); y = at.dmatrix("y"); v = at.dvector("v"); w = at.dvector("w"); t4 = at.dtensor4("t4")
    i8 = at.bmatrix("i8"); u8 = at.TensorType("uint8", (None, None))("u8"); i16 = at.TensorType("int16", (None, None))("i16"); f32 = at.fmatrix("f32")
    n = at.lscalar("n"); s = at.dscalar("s")
    VALS2 = {x: rng.standard_normal((5, 4)), y: rng.standard_normal((5, 4)), v: rng.standard_normal(4), w: rng.standard_normal(5),
            t4: rng.standard_normal((2, 3, 4, 5)), i8: rng.integers(-100, 100, (5, 4)).astype("int8"), u8: rng.integers(0, 255, (5, 4)).astype("uint8"),
            i16: rng.integers(-3000, 3000, (5, 4)).astype("int16"), f32: rng.standard_normal((5, 4)).astype("float32"), n: np.int64(3), s: np.float64(0.7)}
    def scan1(): return ae.scan(lambda a, b, h, W: at.tanh(at.dot(h, W) + a * b), sequences=[x, y], outputs_info=[at.zeros((4,))], non_sequences=[at.dot(x.T, x)])[0]
    def scan_back(): return ae.scan(lambda a, h: h * 0.5 + a, sequences=[x], outputs_info=[at.zeros((4,))], go_backwards=True)[0]
    def scan_nsteps(): return ae.scan(lambda h: h * 1.1 + 1, outputs_info=[v], n_steps=n)[0]
    def scan_taps(): return ae.scan(lambda a, h1, h2: h1 + 0.5 * h2 + a.sum(), sequences=[x], outputs_info=[dict(initial=at.stack([v[0], v[1]]), taps=[-1, -2])])[0]
    def scan_two(): 
        (a, b), _ = ae.scan(lambda r, p, q: (p + r, q * 0.9 + r.sum()), sequences=[x], outputs_info=[at.zeros((4,)), at.as_tensor(1.0)])
        return [a[-1], b]
    def scan_grad(): 
        h, _ = ae.scan(lambda a, h, W: at.tanh(at.dot(h, W) + a), sequences=[x], outputs_info=[at.zeros((4,))], non_sequences=[y[:4]])
        return ae.grad(h[-1].sum(), [x, y])
    def scan_shared():
        c = ae.shared(np.float64(0.0), name="c")
        o, upd = ae.scan(lambda a: (a.sum() + c, {c: c + 1.0}), sequences=[x])
        return o
    E2 = {
     "int8_arith": lambda: (i8 + i8) * 2 - i8 // 3 + i8 % 7, "uint8_arith": lambda: u8 + u8 * 2 + (u8 >> 1 if hasattr(u8, "__rshift__") else u8), "int16_mixed": lambda: i16 * i8 + u8,
     "int_float_mix": lambda: i8 * x + i16 / 3 + f32, "f32_f64_mix": lambda: f32 * x + f32.sum(), "int_sum_prod": lambda: [i8.sum(), i8.sum(axis=0), u8.prod(axis=1), i16.max(), i8.min(axis=1)],
     "int_mean_var": lambda: [i8.mean(), i16.var(axis=0), u8.mean(axis=1)], "bool_sum": lambda: (x > 0).sum(axis=0) + (y > 0).sum(), "cmp_chain": lambda: at.and_(at.gt(x, 0), at.lt(y, 0.5)) | at.eq(i8, 3),
     "t4_reduce": lambda: [t4.sum(axis=(0, 2)), t4.max(axis=(1, 3)), t4.mean(axis=(0, 1, 2)), t4.prod(axis=3).sum(axis=0)], "t4_shuffle": lambda: t4.dimshuffle(3, 1, "x", 0, 2) * 2 + 1,
     "t4_index": lambda: t4[1, :, ::2, -1] + t4[:, 1:, 0, 2:4].sum(), "bcast_rank": lambda: x[None, :, :, None] * t4[:, :1, :, :] if False else x[:, None, :] * v[None, None, :] + w[:, None, None],
     "zero_size": lambda: [x[:0].sum(axis=0), x[:0] * 2, at.dot(x[:, :0], y[:, :0].T), x[:0].sum()], "zero_d": lambda: [s * 2 + x[0, 0], at.exp(s), s ** 2 + v.sum()],
     "outer_ops": lambda: at.outer(w, v) * x + at.dot(w, x) + at.dot(x, v)[:, None], "dot_chain": lambda: at.dot(at.dot(x, y.T), at.dot(x, x.T)) @ w, "gemm_pattern": lambda: 0.3 * x + 0.7 * at.dot(x, at.dot(y.T, y)),
     "gemv_pattern": lambda: 0.5 * w + 2.0 * at.dot(x, v), "ger_pattern": lambda: x + 0.1 * at.outer(w, v), "dot22scalar": lambda: 3.0 * at.dot(x, y.T),
     "grad_tanh_mlp": lambda: ae.grad(at.tanh(at.dot(at.tanh(at.dot(x, y.T)), x)).sum(), [x, y]), "grad_sigmoid_bce": lambda: ae.grad(-(at.log(at.sigmoid(x)) * (y > 0) + at.log(1 - at.sigmoid(x)) * (y <= 0)).mean(), x),
     "grad_prod": lambda: ae.grad(x.prod(axis=1).sum(), x), "grad_abs_sqrt": lambda: ae.grad(at.sqrt(abs(x) + 1).sum() + abs(y).sum(), [x, y]), "grad_minmax": lambda: ae.grad(at.maximum(x, y).sum() + at.minimum(x, 0.1).sum(), [x, y]),
     "grad_softplus_erf": lambda: ae.grad((at.softplus(x) * at.erf(y)).sum(), [x, y]), "grad_logsumexp": lambda: ae.grad(at.logsumexp(x, axis=1).sum(), x), "grad_gammaln_psi": lambda: ae.grad(at.gammaln(abs(x) + 1).sum(), x),
     "grad_dimshuffle_bcast": lambda: ae.grad((x * v[None, :] + w[:, None]).sum() ** 2, [v, w]), "grad_reshape_flatten": lambda: ae.grad((x.flatten()[3:11].reshape((2, 4)) ** 2).sum(), x), "grad_inc_subtensor": lambda: ae.grad(at.set_subtensor(x[1:3], y[1:3] * 2).sum() * 2, [x, y]),
     "grad_advsub": lambda: ae.grad(x[[0, 2, 2, 4]].sum() * 3 + (x[:, [1, 1, 3]] ** 2).sum(), x), "grad_switch_where": lambda: ae.grad(at.switch(x > y, x * x, y * 3).sum(), [x, y]), "grad_mean_std": lambda: ae.grad(((x - x.mean(axis=0)) / (x.std(axis=0) + 1e-3)).sum(axis=1).max(), x),
     "hessian_vec": lambda: ae.grad(ae.grad((at.tanh(x) ** 2).sum(), x).sum(), x), "jacobian": lambda: ae.gradient.jacobian(at.tanh(at.dot(x, v)), v),
     "scan1": scan1, "scan_back": scan_back, "scan_nsteps": scan_nsteps, "scan_taps": scan_taps, "scan_grad": scan_grad, "scan_shared": scan_shared,
     "clip_grad": lambda: ae.grad(at.clip(x, -0.3, 0.3).sum(), x), "pow_int": lambda: x ** 3 + x ** -2 + 2.0 ** y, "mod_float": lambda: at.mod(x, 0.7) + at.true_div(i8, 4), "round_modes": lambda: at.round(x * 3, mode="half_to_even") + at.round(y * 3, mode="half_away_from_zero"),
     "isclose_all": lambda: at.allclose(x, x + 1e-9), "argmax_keep": lambda: [at.argmax(x, axis=1, keepdims=True), at.max_and_argmax(x, axis=0)[1]], "cumsum_none": lambda: at.cumsum(x), "tile_repeat_grad": lambda: ae.grad((at.tile(v, (3, 1)) * x[:3]).sum() + at.repeat(w, 2).sum(), [v, w]),
    }
    return E2, VALS2


def sweep3():
    """tensor.nnet front end"""
    import aesara.tensor.nnet as nn
    rng = np.random.default_rng(0)
    x = at.dmatrix("x"); W = at.dmatrix("W"); b = at.dvector("b"); y = at.lvector("y"); t = at.dmatrix("t")
    vals = {x: rng.standard_normal((6, 5)), W: rng.standard_normal((5, 4)), b: rng.standard_normal(4), y: rng.integers(0, 4, 6), t: rng.random((6, 4))}
    E = {
     "logreg_tutorial": lambda: [nn.categorical_crossentropy(nn.softmax(at.dot(x, W) + b), y).mean()] + ae.grad(nn.categorical_crossentropy(nn.softmax(at.dot(x, W) + b), y).mean(), [W, b]),
     "xent_onehot_dense": lambda: ae.grad(nn.categorical_crossentropy(nn.softmax(at.dot(x, W)), t / t.sum(axis=1, keepdims=True)).sum(), W),
     "neg_log_likelihood": lambda: ae.grad(-at.mean(at.log(nn.softmax(at.dot(x, W) + b))[at.arange(y.shape[0]), y]), [W, b]),
     "binary_xent": lambda: ae.grad(nn.binary_crossentropy(at.sigmoid(at.dot(x, W)), t).mean(), W),
     "relu_elu_softplus": lambda: [nn.relu(x), nn.relu(x, 0.1), nn.elu(x), nn.softsign(x), nn.hard_sigmoid(x)],
     "logsoftmax_nnet": lambda: ae.grad(nn.logsoftmax(at.dot(x, W))[at.arange(6), y].sum(), W),
     "confusion_argmax": lambda: [at.argmax(nn.softmax(at.dot(x, W)), axis=1), at.neq(at.argmax(at.dot(x, W), axis=1), y).mean()],
     "sigmoid_bce_logits": lambda: ae.grad(nn.sigmoid_binary_crossentropy(at.dot(x, W), t).sum(), W) if hasattr(nn, "sigmoid_binary_crossentropy") else x,
     "l2_weight_decay_sgd": lambda: [W - 0.1 * ae.grad(nn.categorical_crossentropy(nn.softmax(at.dot(x, W) + b), y).mean() + 1e-3 * (W ** 2).sum(), W)],
     "batchnorm_like": lambda: ae.grad((((x - x.mean(0)) / at.sqrt(x.var(0) + 1e-5)) * b[:4].sum()).sum() + (x ** 2).sum(), x),
     "dropout_mask_mul": lambda: x * (t[:, :1] > 0.5) / 0.5,
     "onehot_xent": lambda: nn.categorical_crossentropy(nn.softmax(at.dot(x, W)), at.extra_ops.to_one_hot(y, 4)).mean(),
    }
    return E, vals


def sweep4():
    """gradient helpers, extra_ops, odds and ends"""
    from aesara.gradient import grad_clip, disconnected_grad, zero_grad
    rng = np.random.default_rng(0)
    x = at.dmatrix("x"); v = at.dvector("v"); i = at.lvector("i")
    vals = {x: rng.standard_normal((5, 4)), v: rng.standard_normal(4), i: np.array([3, 0, 2, 2])}
    E = {
     "grad_clip": lambda: ae.grad((grad_clip(x, -0.1, 0.1) ** 2).sum(), x),
     "disconnected_zero_grad": lambda: ae.grad((disconnected_grad(x) * x + zero_grad(x.sum()) * x).sum(), x),
     "print_op": lambda: ae.printing.Print("dbg")(x) * 2 if False else x * 2,
     "fill_diagonal": lambda: at.extra_ops.fill_diagonal(x[:4], 7.0),
     "squeeze_bcast": lambda: at.squeeze(x.dimshuffle(0, "x", 1)) + 1,
     "ravel_unravel": lambda: list(at.extra_ops.unravel_index(i, (2, 2))) + [at.extra_ops.ravel_multi_index((i % 2, i // 2), (2, 2))],
     "broadcast_arrays": lambda: at.extra_ops.broadcast_arrays(v, x)[0] * 1,
     "choose": lambda: at.choose(i % 2, [v, v * 2]),
     "minimum_where_nan": lambda: at.where(at.isnan(x / (x - x)), 0.0, x),
     "outer_sub": lambda: v[:, None] - v[None, :],
     "stack_list_scalars": lambda: at.stack([x[0, 0], x[1, 1], v[2]]) * 2,
     "as_tensor_const_big": lambda: x + np.arange(20.0).reshape(5, 4),
     "const_big_5000": lambda: v[0] + at.as_tensor(np.arange(5000.0)).sum(),
     "mean_dtype": lambda: [x.mean(dtype="float32"), x.sum(dtype="float32", axis=0), at.cast(x, "int32").mean()],
     "pow_grad": lambda: ae.grad((abs(x) ** v).sum(), [x, v]),
     "second_fill_like": lambda: at.fill(x, v[0]) + at.zeros_like(x, dtype="int32"),
     "isfinite_any": lambda: at.any(at.isinf(x)) | at.all(at.isnan(x)),
     "shape_of_shape": lambda: at.as_tensor(x.shape).sum() + x.shape[0] * x.ndim,
     "set_sub_scalar_idx": lambda: at.set_subtensor(x[i[0], i[1]], 5.0),
     "inc_sub_neg_step": lambda: at.inc_subtensor(x[::-1, ::-2], 1.0),
     "tensordot_axes2": lambda: at.tensordot(x, x, axes=2),
     "dot_vec_vec": lambda: at.dot(v, v) + at.dot(x, v).sum(),
     "rsub_rdiv_scalars": lambda: (1 - x) / (2 / (x + 3)),
     "neg_pow_int": lambda: (-x) ** 2 + (x ** 2) ** 0.5,
     "int_true_div": lambda: i / 2 + i // 2 + at.true_div(i, i + 1),
     "clip_int": lambda: at.clip(i, 1, 2),
     "max_keepdims_sub": lambda: at.exp(x - x.max(axis=1, keepdims=True)).sum(axis=1),
     "flatten_ndim": lambda: at.flatten(x.dimshuffle(0, 1, "x"), ndim=2) * 1,
     "join_vectors_scalar": lambda: at.concatenate([v, at.stack([x.sum()]), v[:2]]),
     "alloc_broadcast_grad": lambda: ae.grad((at.alloc(v, 3, 4) * x[:3]).sum(), v),
    }
    return E, vals


def run(exprs, vals):
    ok, bad = [], []
    ins = list(vals)
    for name, mk in exprs.items():
        try:
            out = mk()
            outs = list(out) if isinstance(out, (list, tuple)) else [out]
            linker = HipLinker(executor_factory=lambda plan: (lambda *a: interp.run_plan(plan, a)))
            f = ae.function(ins, outs, mode=Mode(linker, HIP_QUERY), on_unused_input="ignore")
            got = f(*[vals[k] for k in ins])
            try:
                fr = ae.function(ins, outs, mode=Mode("cvm", "fast_run"), on_unused_input="ignore")
                want = fr(*[vals[k] for k in ins])
            except Exception:  # noqa: BLE001  a reference C thunk that does not build on NumPy 2
                fr = ae.function(ins, outs, mode=Mode("py", "fast_run"), on_unused_input="ignore")
                want = fr(*[vals[k] for k in ins])
            match = all(np.asarray(g).shape == np.asarray(w_).shape and
                        np.allclose(np.asarray(g), np.asarray(w_), rtol=1e-6, atol=1e-8, equal_nan=True)
                        for g, w_ in zip(got, want))
            (ok if match else bad).append((name, "" if match else "MISMATCH"))
        except Exception as e:  # noqa: BLE001
            bad.append((name, type(e).__name__ + ": " + str(e).split("\n")[0][:140]))
    return ok, bad


if __name__ == "__main__":
    total_ok = 0
    lines = []
    for label, (exprs, vals) in (("tensor API", sweep1()), ("dtypes / gradients / scans", sweep2()),
                                 ("tensor.nnet front end", sweep3()), ("gradient helpers / extra_ops", sweep4())):
        ok, bad = run(exprs, vals)
        total_ok += len(ok)
        lines.append("== %s: %d of %d match the reference (C linker; Python linker where a C thunk does not build here)" % (label, len(ok), len(exprs)))
        lines.append("   ok: " + " ".join(n_ for n_, _ in ok))
        for n_, m in bad:
            lines.append("   not lowered / failed: %-22s %s" % (n_, m))
    print("\n".join(lines))
