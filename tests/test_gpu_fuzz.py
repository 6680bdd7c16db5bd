"""Randomised parity of the generated Elemwise / CAReduce kernels against the oracle: random ranks,
broadcast patterns, operand views (transposed / stepped / reversed), dtypes and scalar programs,
optionally followed by a CAReduce over random axes.  Plans are built by hand (no reference needed
on the GPU box); the oracle (oracle/interp.py) is the checker.  Integer / bool results bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLOATS = ("float32", "float64")
INTS = ("int8", "int32", "int64")


def _rand_shape(rng, nd):
    return [int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64])) for _ in range(nd)]


def _rand_array(rng, shape, dtype):
    if dtype == "bool":
        return rng.random(shape) < 0.5
    if dtype in INTS:
        lo, hi = (-100, 100) if dtype == "int8" else (-1000, 1000)
        return rng.integers(lo, hi, shape).astype(dtype)
    return rng.standard_normal(shape).astype(dtype)


def _rand_view(rng, a):
    """A random non-contiguous presentation of the same values (base buffer + view)."""
    kind = rng.integers(4)
    if a.ndim == 0 or kind == 0:
        return a
    if kind == 1 and a.ndim >= 2:                      # transposed storage
        perm = rng.permutation(a.ndim)
        base = np.ascontiguousarray(a.transpose(perm))
        return base.transpose(np.argsort(perm))
    if kind == 2:                                      # stepped storage
        base = np.zeros([2 * s for s in a.shape], a.dtype)
        v = base[tuple(slice(None, None, 2) for _ in a.shape)]
        v[...] = a
        return v
    base = np.ascontiguousarray(a[tuple(slice(None, None, -1) for _ in a.shape)])
    return base[tuple(slice(None, None, -1) for _ in a.shape)]   # reversed storage


def _rand_program(rng, dtypes, out_dtype):
    """A random scalar expression over the inputs that stays finite / in range."""
    n = len(dtypes)
    nodes, cur = [], []
    isf = out_dtype in FLOATS
    for k, dt in enumerate(dtypes):
        nodes.append({"op": "cast", "in": [["i", k]], "dtype": out_dtype})
        cur.append(["t", len(nodes) - 1])
    ops_f = ["add", "mul", "sub", "maximum", "minimum", "tanh", "neg", "abs", "sqr", "sigmoid", "switch"]
    ops_i = ["add", "sub", "maximum", "minimum", "neg", "abs", "switch", "and", "or", "xor"]
    for _ in range(int(rng.integers(1, 6))):
        op = str(rng.choice(ops_f if isf else ops_i))
        a, b = cur[int(rng.integers(len(cur)))], cur[int(rng.integers(len(cur)))]
        if op in ("tanh", "neg", "abs", "sqr", "sigmoid"):
            if op == "sqr":      # keep magnitudes bounded
                nodes.append({"op": "tanh" if isf else "neg", "in": [a], "dtype": out_dtype})
                a = ["t", len(nodes) - 1]
            nodes.append({"op": op, "in": [a], "dtype": out_dtype})
        elif op == "switch":
            nodes.append({"op": "gt", "in": [a, b], "dtype": "bool"})
            nodes.append({"op": "switch", "in": [["t", len(nodes) - 1], b, a], "dtype": out_dtype})
        elif op == "mul":
            nodes.append({"op": "tanh", "in": [a], "dtype": out_dtype})
            nodes.append({"op": "mul", "in": [["t", len(nodes) - 1], b], "dtype": out_dtype})
        else:
            nodes.append({"op": op, "in": [a, b], "dtype": out_dtype})
        cur.append(["t", len(nodes) - 1])
    return {"n_in": n, "nodes": nodes, "out": [cur[-1]]}


def _to_dev(a):
    import torch
    from aesara_amd.device import DevArray
    base = a
    while base.base is not None and isinstance(base.base, np.ndarray):
        base = base.base
    if base is a or a.size == 0 or any(s < 0 for s in a.strides):
        if any(s < 0 for s in a.strides) and a.size:
            # negative strides: upload the base, address the view from its first element
            d = DevArray.from_numpy(np.ascontiguousarray(base), torch.device("cuda"))
            off = (a.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // a.itemsize
            return d.view(a.shape, [s // a.itemsize for s in a.strides], off)
        return DevArray.from_numpy(np.ascontiguousarray(a), torch.device("cuda"))
    d = DevArray.from_numpy(np.ascontiguousarray(base), torch.device("cuda"))
    off = (a.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // a.itemsize
    return d.view(a.shape, [s // a.itemsize for s in a.strides], off)


@pytest.mark.parametrize("seed", range(6))
def test_random_elemwise_and_reduce_plans_match_the_oracle(seed):
    import interp
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan, Var
    rng = np.random.default_rng(1000 + seed)
    for trial in range(30):
        nd = int(rng.integers(1, 5))
        full = _rand_shape(rng, nd)
        out_dtype = str(rng.choice(FLOATS + INTS[1:]))
        nin = int(rng.integers(1, 4))
        pool = (FLOATS + INTS + ("bool",)) if out_dtype in FLOATS else (INTS[:2] if out_dtype == "int32" else INTS) + ("bool",)
        dtypes = [str(rng.choice(pool)) for _ in range(nin)]
        if out_dtype == "float32":
            dtypes = [d if d != "float64" and d != "int64" else "float32" for d in dtypes]
        shapes = [[(1 if rng.random() < 0.3 else s) for s in full] for _ in range(nin)]
        shapes[0] = list(full)
        arrays = [_rand_view(rng, _rand_array(rng, sh, dt)) for sh, dt in zip(shapes, dtypes)]
        prog = _rand_program(rng, dtypes, out_dtype)
        vs = {k: Var(k, dtypes[k], [1 if s == 1 else None for s in shapes[k]]) for k in range(nin)}
        vs[nin] = Var(nin, out_dtype, [None] * nd)
        nodes = [Node("Elemwise", list(range(nin)), [nin], {"scalar": prog})]
        outs = [nin]
        if rng.random() < 0.7:
            axes = sorted(set(int(a) for a in rng.integers(0, nd, int(rng.integers(1, nd + 1)))))
            axis = None if rng.random() < 0.25 else axes
            red_op = str(rng.choice(["add", "maximum", "minimum"] if out_dtype in FLOATS
                                    else ["add", "maximum", "minimum", "and", "or", "xor"]))
            kept = 0 if axis is None else nd - len(axes)
            acc = "float64" if (out_dtype in FLOATS and red_op == "add") else out_dtype
            vs[nin + 1] = Var(nin + 1, out_dtype, [None] * kept)
            nodes.append(Node("CAReduce", [nin], [nin + 1], {"scalar_op": red_op, "axis": axis,
                                                              "acc_dtype": acc}))
            outs = [nin + 1] if rng.random() < 0.5 else [nin, nin + 1]
        plan = Plan("fuzz", vs, list(range(nin)), outs, nodes)
        want = interp.run_plan(plan, [np.asarray(a) for a in arrays])
        for fuse in (True, False):
            got = PlanExecutor(plan, fuse=fuse)(*[_to_dev(a) for a in arrays])
            for g, w in zip(got, want):
                g = g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g)
                ctx = (seed, trial, fuse, full, dtypes, out_dtype, plan.pretty())
                assert g.shape == np.shape(w) and g.dtype == np.asarray(w).dtype, ctx
                if out_dtype in FLOATS:
                    tol = 2e-5 if out_dtype == "float32" else 1e-11
                    np.testing.assert_allclose(g, w, rtol=tol, atol=tol * max(1.0, float(np.abs(w).max(initial=0))),
                                               err_msg=str(ctx))
                else:
                    assert np.array_equal(g, w), ctx


def _rand_basic_index(rng, shape):
    """Random basic index (ints and slices, negative / out-of-range bounds and steps included) and
    its plan encoding (constants inline, some entries dynamic)."""
    idx, enc, dyn = [], [], []

    def maybe_dyn(v):
        if v is not None and rng.random() < 0.4:
            dyn.append(np.int64(v))
            return "in"
        return v
    for n in shape[:int(rng.integers(1, len(shape) + 1))]:
        if rng.random() < 0.3 and n > 0:
            i = int(rng.integers(-n, n))
            idx.append(i)
            enc.append({"index": maybe_dyn(i)})
        else:
            def bound():
                return None if rng.random() < 0.3 else int(rng.integers(-n - 3, n + 4))
            st, sp = bound(), bound()
            se = None if rng.random() < 0.4 else int(rng.choice([-3, -2, -1, 1, 2, 3]))
            idx.append(slice(st, sp, se))
            enc.append({"slice": [maybe_dyn(st), maybe_dyn(sp), maybe_dyn(se)]})
    return tuple(idx), enc, dyn


@pytest.mark.parametrize("seed", range(3))
def test_random_indexing_plans_are_bit_exact(seed):
    """Subtensor views, IncSubtensor (set / inc, broadcast values), AdvancedSubtensor1 /
    AdvancedIncSubtensor1 and mixed advanced indices with random bounds, steps, negative and
    repeated indices: bit-exact against NumPy indexing (through the oracle)."""
    import interp
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan, Var
    rng = np.random.default_rng(2000 + seed)
    ident = {"n_in": 1, "nodes": [{"op": "neg", "in": [["i", 0]], "dtype": "int64"}], "out": [["t", 0]]}
    for trial in range(40):
        nd = int(rng.integers(1, 4))
        shape = [int(rng.integers(1, 9)) for _ in range(nd)]
        x = rng.integers(-1000, 1000, shape).astype("int64")
        index, enc, dyn = _rand_basic_index(rng, shape)
        sub = x[index]
        vs = {0: Var(0, "int64", [None] * nd)}
        for k in range(len(dyn)):
            vs[1 + k] = Var(1 + k, "int64", [])
        b = 1 + len(dyn)
        dyn_ids = list(range(1, b))
        vs[b] = Var(b, "int64", [None] * sub.ndim)              # the view
        vs[b + 1] = Var(b + 1, "int64", [None] * sub.ndim)      # -view (forces a kernel on it)
        yshape = [s if rng.random() < 0.6 else 1 for s in sub.shape][int(rng.integers(0, sub.ndim + 1)):]
        y = rng.integers(-50, 50, yshape).astype("int64")
        vs[b + 2] = Var(b + 2, "int64", [1 if s == 1 else None for s in yshape])
        vs[b + 3] = Var(b + 3, "int64", [None] * nd)
        vs[b + 4] = Var(b + 4, "int64", [None] * nd)
        nodes = [Node("Subtensor", [0] + dyn_ids, [b], {"idx_list": enc}),
                 Node("Elemwise", [b], [b + 1], {"scalar": ident}),
                 Node("IncSubtensor", [0, b + 2] + dyn_ids, [b + 3],
                      {"idx_list": enc, "set_instead_of_inc": True, "inplace": False}),
                 Node("IncSubtensor", [0, b + 2] + dyn_ids, [b + 4],
                      {"idx_list": enc, "set_instead_of_inc": False, "inplace": False})]
        ins_ids = [0] + dyn_ids + [b + 2]
        plan = Plan("fuzz_idx", vs, ins_ids, [b + 1, b + 3, b + 4], nodes)
        args = [x] + dyn + [y]
        want = interp.run_plan(plan, args)
        got = PlanExecutor(plan)(*[_to_dev(a) if np.ndim(a) else a for a in args])
        for g, w in zip(got, want):
            g = g.cpu().numpy() if hasattr(g, "cpu") else np.asarray(g)
            assert g.shape == np.shape(w) and np.array_equal(g, w), (seed, trial, shape, index, yshape)
        # integer-vector rows: gather, scatter-set (unique rows), scatter-add (repeats)
        rows = int(rng.integers(2, 12))
        m = rng.integers(-1000, 1000, (rows, int(rng.integers(1, 6)))).astype("int64")
        k = int(rng.integers(1, 15))
        idx = rng.integers(-rows, rows, k).astype(str(rng.choice(["int64", "int32", "int8"])))
        uniq = rng.permutation(rows)[:min(k, rows)].astype("int64")
        yv = rng.integers(-9, 9, (k, m.shape[1])).astype("int64")
        vs = {0: Var(0, "int64", [None, None]), 1: Var(1, idx.dtype.name, [None]), 2: Var(2, "int64", [None, None]),
              3: Var(3, "int64", [None]), 4: Var(4, "int64", [None, None]), 5: Var(5, "int64", [None, None]),
              6: Var(6, "int64", [None, None])}
        nodes = [Node("AdvancedSubtensor1", [0, 1], [4], {}),
                 Node("AdvancedIncSubtensor1", [0, 2, 1], [5], {"set_instead_of_inc": False, "inplace": False}),
                 Node("AdvancedIncSubtensor1", [0, 4, 3], [6], {"set_instead_of_inc": True, "inplace": False})]
        # output 6: x[uniq] = x[idx][:len(uniq)] — build y for the set from the gather
        plan = Plan("fuzz_adv1", vs, [0, 1, 2, 3], [4, 5], nodes[:2])
        args = [m, idx, yv, uniq]
        want = interp.run_plan(plan, args)
        got = PlanExecutor(plan)(*[_to_dev(a) for a in args])
        for g, w in zip(got, want):
            assert np.array_equal(g.cpu().numpy(), w), (seed, trial, "adv1", rows, idx)
        ys = rng.integers(-9, 9, (len(uniq), m.shape[1])).astype("int64")
        vs2 = {0: Var(0, "int64", [None, None]), 1: Var(1, "int64", [None, None]), 2: Var(2, "int64", [None]),
               3: Var(3, "int64", [None, None])}
        plan = Plan("fuzz_set1", vs2, [0, 1, 2], [3], [Node("AdvancedIncSubtensor1", [0, 1, 2], [3],
                                                            {"set_instead_of_inc": True, "inplace": False})])
        (want,) = interp.run_plan(plan, [m, ys, uniq])
        (got,) = PlanExecutor(plan)(_to_dev(m), _to_dev(ys), _to_dev(uniq))
        assert np.array_equal(got.cpu().numpy(), want), (seed, trial, "set1")
        # mixed advanced index on a 3-d array: arrays / slices / newaxis in random positions
        t3 = rng.integers(-1000, 1000, [int(rng.integers(2, 6)) for _ in range(3)]).astype("int64")
        n_arr = int(rng.integers(1, 3))
        pos = sorted(rng.permutation(3)[:n_arr].tolist())
        L = int(rng.integers(1, 5))
        entries, extra, np_idx = [], [], []
        for d in range(3):
            if d in pos:
                ia = rng.integers(-t3.shape[d], t3.shape[d], L).astype("int64")
                entries.append({"array": len(extra)})
                extra.append(ia)
                np_idx.append(ia)
            elif d < max(pos) or rng.random() < 0.7:
                se = int(rng.choice([-2, -1, 1, 2]))
                entries.append({"slice": [None, None, se]})
                np_idx.append(slice(None, None, se))
            else:
                break
            if rng.random() < 0.15:
                entries.append({"newaxis": True})
                np_idx.append(None)
        res = t3[tuple(np_idx)]
        vs3 = {0: Var(0, "int64", [None] * 3)}
        for q in range(len(extra)):
            vs3[1 + q] = Var(1 + q, "int64", [None])
        o = 1 + len(extra)
        vs3[o] = Var(o, "int64", [None] * res.ndim)
        plan = Plan("fuzz_advmix", vs3, list(range(o)), [o],
                    [Node("AdvancedSubtensor", list(range(o)), [o], {"index": entries})])
        (want,) = interp.run_plan(plan, [t3] + extra)
        assert np.array_equal(want, res)
        (got,) = PlanExecutor(plan)(*[_to_dev(a) for a in [t3] + extra])
        assert np.array_equal(got.cpu().numpy(), res), (seed, trial, "advmix", t3.shape, entries)


@pytest.mark.parametrize("seed", range(2))
def test_random_blas_shapes_and_layouts(seed):
    """Gemm / Gemv / Ger / BatchedDot with random extents (0, 1, ragged, around the tile edges),
    operand layouts (row-major, transposed, stepped views) and alpha / beta, against an fp64
    NumPy restatement: Frobenius-relative 2e-6 in fp32 (the bar of BASELINE config 3b), 1e-13 fp64."""
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan, Var
    rng = np.random.default_rng(3000 + seed)
    sizes = [0, 1, 2, 3, 7, 16, 17, 31, 33, 64, 65, 100, 128, 129, 200, 257]

    def close(got, want, dt):
        got = got.cpu().numpy().astype("float64")
        tol = 2e-6 if dt == "float32" else 1e-13
        denom = max(np.linalg.norm(want), 1e-30)
        return got.shape == want.shape and (want.size == 0 or np.linalg.norm(got - want) / denom <= tol
                                            or np.abs(got - want).max() <= tol)

    def mat(shape, dt):
        return _rand_view(rng, rng.standard_normal(shape).astype(dt))
    for trial in range(40):
        dt = str(rng.choice(FLOATS))
        M, N, K = (int(rng.choice(sizes)) for _ in range(3))
        al, be = float(rng.choice([1.0, 0.8, -1.5])), float(rng.choice([0.0, 1.0, 0.4]))
        z, x, y = mat((M, N), dt), mat((M, K), dt), mat((K, N), dt)
        v2 = [Var(0, dt, [None, None]), Var(1, dt, []), Var(2, dt, [None, None]), Var(3, dt, [None, None]),
              Var(4, dt, []), Var(5, dt, [None, None])]
        plan = Plan("fz_gemm", {v.id: v for v in v2}, [0, 1, 2, 3, 4], [5],
                    [Node("Gemm", [0, 1, 2, 3, 4], [5], {"inplace": False})])
        (got,) = PlanExecutor(plan)(_to_dev(z), np.asarray(al, dt), _to_dev(x), _to_dev(y), np.asarray(be, dt))
        want = be * z.astype("float64") + al * (x.astype("float64") @ y.astype("float64"))
        assert close(got, want, dt), ("gemm", seed, trial, dt, M, N, K, al, be)
        # Gemv: y <- beta y + alpha A x
        yv, xv = rng.standard_normal(M).astype(dt), rng.standard_normal(K).astype(dt)
        if rng.random() < 0.5 and K:
            xv = np.ascontiguousarray(np.repeat(xv, 2))[::2]
        vv = [Var(0, dt, [None]), Var(1, dt, []), Var(2, dt, [None, None]), Var(3, dt, [None]), Var(4, dt, []),
              Var(5, dt, [None])]
        plan = Plan("fz_gemv", {v.id: v for v in vv}, [0, 1, 2, 3, 4], [5],
                    [Node("Gemv", [0, 1, 2, 3, 4], [5], {"inplace": False})])
        (got,) = PlanExecutor(plan)(_to_dev(yv), np.asarray(al, dt), _to_dev(x), _to_dev(xv), np.asarray(be, dt))
        want = be * yv.astype("float64") + al * (x.astype("float64") @ xv.astype("float64"))
        assert close(got, want, dt), ("gemv", seed, trial, dt, M, K, al, be)
        # Ger: A + alpha outer(u, w)
        u, w = rng.standard_normal(M).astype(dt), rng.standard_normal(N).astype(dt)
        vg = [Var(0, dt, [None, None]), Var(1, dt, []), Var(2, dt, [None]), Var(3, dt, [None]), Var(4, dt, [None, None])]
        plan = Plan("fz_ger", {v.id: v for v in vg}, [0, 1, 2, 3], [4],
                    [Node("Ger", [0, 1, 2, 3], [4], {"destructive": False})])
        (got,) = PlanExecutor(plan)(_to_dev(z), np.asarray(al, dt), _to_dev(u), _to_dev(w))
        want = z.astype("float64") + al * np.outer(u.astype("float64"), w.astype("float64"))
        assert close(got, want, dt), ("ger", seed, trial, dt, M, N)
        # BatchedDot
        B = int(rng.choice([1, 2, 5]))
        m2, n2, k2 = (int(rng.choice(sizes[1:10])) for _ in range(3))
        a3, b3 = mat((B, m2, k2), dt), mat((B, k2, n2), dt)
        vb = [Var(0, dt, [None] * 3), Var(1, dt, [None] * 3), Var(2, dt, [None] * 3)]
        plan = Plan("fz_bdot", {v.id: v for v in vb}, [0, 1], [2], [Node("BatchedDot", [0, 1], [2], {})])
        (got,) = PlanExecutor(plan)(_to_dev(a3), _to_dev(b3))
        assert close(got, a3.astype("float64") @ b3.astype("float64"), dt), ("bdot", seed, trial, dt, B, m2, n2, k2)


def test_row_chains_and_gemm_epilogues_on_operand_views():
    """The fused row-chain / GEMM-epilogue / GEMV-chain kernels with transposed, stepped and
    reversed operand storage (they must fall back or address the views correctly): fused ==
    unfused == oracle on random shapes."""
    import interp
    from golden_util import CASES, case_plan
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(77)

    def plan_of(name):
        return case_plan(next(c for c in CASES if c["name"] == name))
    sizes = [1, 2, 5, 16, 17, 33, 64, 100, 129]
    for name, tol in (("softmax_rows_f32", 3e-6), ("logsoftmax_rows_f64", 1e-12)):
        dt = "float32" if name.endswith("f32") else "float64"
        for _ in range(12):
            n, k = int(rng.choice(sizes)), int(rng.choice(sizes[1:]))
            x = _rand_view(rng, (rng.standard_normal((n, k)) * 3).astype(dt))
            (want,) = interp.run_plan(plan_of(name), [np.asarray(x)])
            for fuse in (True, False):
                (got,) = PlanExecutor(plan_of(name), fuse=fuse)(_to_dev(x))
                np.testing.assert_allclose(got.cpu().numpy(), want, rtol=tol, atol=tol, err_msg=str((name, n, k, fuse)))
    for _ in range(10):
        a_, b_, k = int(rng.choice(sizes[:6])), int(rng.choice(sizes[:6])), int(rng.choice(sizes[1:]))
        x = _rand_view(rng, (rng.standard_normal((a_, b_, k)) * 2).astype("float32"))
        g, bb = _rand_view(rng, rng.standard_normal(k).astype("float32")), rng.standard_normal(k).astype("float32")
        want = interp.run_plan(plan_of("layernorm_float32"), [np.asarray(x), np.asarray(g), bb])
        got = PlanExecutor(plan_of("layernorm_float32"))(_to_dev(x), _to_dev(g), _to_dev(bb))
        for u, w in zip(got, want):
            np.testing.assert_allclose(u.cpu().numpy(), w, rtol=5e-5, atol=5e-5, err_msg=str((a_, b_, k)))
    # small-batch layers: x[m,k1] y[m,k2] W,W2[k1,n] U[k2,n] Wt[n,k1] b[n]
    for _ in range(12):
        m, k1, k2, n = (int(rng.choice(sizes[:8])) for _ in range(4))
        mk = lambda *sh: _rand_view(rng, (rng.standard_normal(sh) * 0.3).astype("float32"))  # noqa: E731
        args = [mk(m, k1), mk(m, k2), mk(k1, n), mk(k1, n), mk(k2, n), mk(n, k1), mk(n)]
        want = interp.run_plan(plan_of("mlp_layers_float32"), [np.asarray(a) for a in args])
        got = PlanExecutor(plan_of("mlp_layers_float32"))(*[_to_dev(a) for a in args])
        for u, w in zip(got, want):
            np.testing.assert_allclose(u.cpu().numpy(), w, rtol=3e-5, atol=3e-5, err_msg=str((m, k1, k2, n)))
    # GLM pattern (config 5) and GEMV chains on views
    for _ in range(8):
        n, d = int(rng.choice([64, 100, 257, 1024])), int(rng.choice([16, 64, 100, 256]))
        X = _rand_view(rng, rng.standard_normal((n, d)).astype("float32"))
        w = _rand_view(rng, (rng.standard_normal(d) / 8).astype("float32"))
        y = (rng.random(n) < 0.5).astype("float32")
        args = [X, w, np.asarray(0.1, "float32"), y]
        want = interp.run_plan(plan_of("cfg5_logistic"), [np.asarray(a) for a in args])
        got = PlanExecutor(plan_of("cfg5_logistic"))(*[_to_dev(a) if np.ndim(a) else a for a in args])
        for u, w_ in zip(got, want):
            u = u.cpu().numpy() if hasattr(u, "cpu") else np.asarray(u)
            np.testing.assert_allclose(u, w_, rtol=5e-5, atol=2e-3, err_msg=str((n, d)))


def _scatter_plan(dtype, idx_dtype, nd):
    from aesara_amd.plan import Node, Plan, Var
    vs = {0: Var(0, dtype, [None] * nd), 1: Var(1, dtype, [None] * nd), 2: Var(2, idx_dtype, [None]),
          3: Var(3, dtype, [None] * nd)}
    return Plan("fuzz_addat", vs, [0, 1, 2], [3],
                [Node("AdvancedIncSubtensor1", [0, 1, 2], [3], {"set_instead_of_inc": False, "inplace": False})])


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_float_scatter_add_is_bit_exact_in_list_order(dtype):
    """AdvancedIncSubtensor1 (inc) on floats: np.add.at adds the contributions of a row in the order
    of the index list (tensor/subtensor.py:2128); the ordered kernel must reproduce that sum bit for
    bit whenever no row receives more than 64 entries — wide vector rows, ragged rows, narrow rows
    and vectors, negative / strided / small-dtype indices, eagerly and replayed."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(77)
    cases = [(8192, 1024, 65536), (1000, 1000, 9000), (513, 257, 4000), (300, 36, 2500), (64, 33, 700),
             (4096, 5, 20000), (50000, 1, 200000), (150000, 2, 100000), (97, 64, 97 * 40), (2, 4096, 100), (1, 128, 64)]
    for rows, width, n in cases:
        shape = (rows, width) if width > 1 else (rows,)
        x = (rng.standard_normal(shape) * 100).astype(dtype)
        y = (rng.standard_normal((n,) + shape[1:]) * 100).astype(dtype)
        if rows * 64 < n:
            n = rows * 64
            y = y[:n]
        # at most 64 hits per row: a random multiset drawn without exceeding the cap
        pool = np.repeat(np.arange(rows), 64)
        idx = rng.permutation(pool)[:n].astype("int64")
        idx = np.where(rng.random(n) < 0.3, idx - rows, idx)          # some negative spellings
        want = x.copy()
        np.add.at(want, idx, y)
        for idt in ("int64", "int32"):
            for use_graph in (False, True):
                ex = PlanExecutor(_scatter_plan(dtype, idt, len(shape)), use_graph=use_graph)
                for _ in range(2 if use_graph else 1):
                    (got,) = ex(_to_dev(x), _to_dev(y), _to_dev(idx.astype(idt)))
                assert np.array_equal(got.cpu().numpy(), want), (rows, width, n, idt, use_graph)
        # a strided index vector (every other entry of a longer one)
        long_idx = np.zeros(2 * n, "int64")
        long_idx[::2] = idx
        (got,) = PlanExecutor(_scatter_plan(dtype, "int64", len(shape)))(
            _to_dev(x), _to_dev(y), torch.from_numpy(long_idx).cuda()[::2])
        assert np.array_equal(got.cpu().numpy(), want), (rows, width, "strided")


def test_float_scatter_add_hot_rows_and_errors():
    """Rows that collect more than 64 entries take the atomic form (close, not bit-exact); the rest
    of the same call stays exact; an out-of-range index raises IndexError like NumPy."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(78)
    rows, width, n = 512, 256, 40000
    x = rng.standard_normal((rows, width)).astype("float32")
    y = rng.standard_normal((n, width)).astype("float32")
    idx = rng.integers(8, rows, n).astype("int64")
    idx[rng.random(n) < 0.3] = 3                                       # one very hot row
    idx[:5000][rng.random(5000) < 0.5] = 5
    want = x.copy()
    np.add.at(want, idx, y)
    ex = PlanExecutor(_scatter_plan("float32", "int64", 2))
    (got,) = ex(_to_dev(x), _to_dev(y), _to_dev(idx))
    got = got.cpu().numpy()
    counts = np.bincount(idx, minlength=rows)
    cold = counts <= 64
    assert cold.sum() > 100 and (~cold).sum() >= 2
    assert np.array_equal(got[cold], want[cold])
    assert np.allclose(got[~cold], want[~cold], rtol=1e-4, atol=1e-2)
    # all entries on one row, and a single-row destination
    one = np.zeros(n, "int64")
    w1 = x[:1].copy()
    np.add.at(w1, one, y)
    (g1,) = ex(_to_dev(x[:1]), _to_dev(y), _to_dev(one))
    assert np.allclose(g1.cpu().numpy(), w1, rtol=1e-4, atol=5e-2)
    # more destination rows than the one-workgroup offset scan takes (ahip_cumulative + hot count)
    big = 140000
    xb = rng.standard_normal((big, 4)).astype("float32")
    ib = rng.integers(100, big, 30000).astype("int64")
    ib[:200] = 7
    yb = rng.standard_normal((30000, 4)).astype("float32")
    wb = xb.copy()
    np.add.at(wb, ib, yb)
    (gb,) = ex(_to_dev(xb), _to_dev(yb), _to_dev(ib))
    gb = gb.cpu().numpy()
    assert np.array_equal(np.delete(gb, 7, 0), np.delete(wb, 7, 0))
    assert np.allclose(gb[7], wb[7], rtol=1e-4, atol=1e-3)
    bad = idx.copy()
    bad[1234] = rows
    with pytest.raises(IndexError):
        (g,) = ex(_to_dev(x), _to_dev(y), _to_dev(bad))
        torch.cuda.synchronize()
        ex.check()


def _rand_dag(rng, R, C, n_ops):
    """A random multi-node plan over float64 matrices holding small integers (every op below is
    exact in fp64, so results compare bit for bit): Elemwise, transposes, row-slice views,
    IncSubtensor set / inc, row gathers and scatter-set / -add, axis sums, Dot22.  Views, values
    read after an update of their base, duplicated and passed-through outputs are all likely."""
    from aesara_amd.plan import Node, Plan, Var
    vs, nodes, shapes = {}, [], {}

    def new(shape, dtype="float64", const=None):
        vid = len(vs)
        vs[vid] = Var(vid, dtype, [None] * len(shape) if const is None else list(shape), None, const)
        shapes[vid] = tuple(shape)
        return vid

    def ew(op, ins):
        nd = {"n_in": len(ins), "nodes": [{"op": op, "in": [["i", k] for k in range(len(ins))],
                                            "dtype": "float64"}], "out": [["t", 0]]}
        o = new(shapes[ins[0]])
        nodes.append(Node("Elemwise", list(ins), [o], {"scalar": nd}))
        return o

    def cint(v):
        return new((), "int64", {"shape": [], "data": [int(v)]})

    ins = [new((R, C)), new((R, C)), new((C, R))]
    idx_in = new((5,), "int64")
    inputs = ins + [idx_in]
    for _ in range(n_ops):
        mats = [v for v, s in shapes.items() if len(s) == 2 and vs[v].dtype == "float64" and min(s) > 0
                and 1 not in vs[v].shape]
        a = int(rng.choice(mats))
        kind = rng.choice(["add", "neg", "T", "rows", "inc", "take", "scat", "sum", "dot", "sub",
                           "center", "bias"])
        same = [v for v in mats if shapes[v] == shapes[a]]
        if kind in ("add", "sub"):
            ew(kind, [a, int(rng.choice(same))])
        elif kind == "neg":
            ew("neg", [a])
        elif kind == "T":
            o = new(shapes[a][::-1])
            nodes.append(Node("DimShuffle", [a], [o], {"new_order": [1, 0]}))
        elif kind == "rows":
            n = shapes[a][0]
            lo, hi = sorted(int(v) for v in rng.integers(0, n + 1, 2))
            st = int(rng.choice([1, 1, 2, -1]))
            sl = slice(lo, hi, st) if st > 0 else slice(hi - 1 if hi else None, lo - 1 if lo else None, st)
            m = len(range(*sl.indices(n)))
            o = new((m, shapes[a][1]))
            nodes.append(Node("Subtensor", [a], [o], {"idx_list": [{"slice": [sl.start, sl.stop, sl.step]}]}))
        elif kind == "inc":
            n = shapes[a][0]
            lo, hi = sorted(int(v) for v in rng.integers(0, n + 1, 2))
            enc = {"idx_list": [{"slice": [lo, hi, None]}]}
            b = int(rng.choice(same))
            y = new((hi - lo, shapes[a][1]))
            nodes.append(Node("Subtensor", [b], [y], dict(enc)))
            o = new(shapes[a])
            nodes.append(Node("IncSubtensor", [a, y], [o], dict(enc, set_instead_of_inc=bool(rng.random() < 0.5),
                                                               inplace=False)))
        elif kind == "take" and shapes[a][0] == R:      # the index vector is drawn for R rows
            o = new((5, shapes[a][1]))
            nodes.append(Node("AdvancedSubtensor1", [a, idx_in], [o], {}))
        elif kind == "scat" and shapes[a][0] == R:
            src = [v for v in mats if shapes[v] == (5, shapes[a][1])]
            if not src:
                continue
            o = new(shapes[a])
            nodes.append(Node("AdvancedIncSubtensor1", [a, int(rng.choice(src)), idx_in], [o],
                              {"set_instead_of_inc": bool(rng.random() < 0.4), "inplace": False}))
        elif kind in ("center", "bias"):
            # m - m.sum(axis, keepdims=True) (a row / column chain) or m + another matrix's sums
            ax = int(rng.integers(0, 2))
            b = a if kind == "center" else int(rng.choice([v for v in mats if shapes[v][1 - ax] == shapes[a][1 - ax]]))
            red = new((shapes[b][1 - ax],))
            nodes.append(Node("CAReduce", [b], [red], {"scalar_op": "add", "axis": [ax], "acc_dtype": "float64"}))
            kshape = [1, shapes[a][1]] if ax == 0 else [shapes[a][0], 1]
            kd = len(vs)
            vs[kd] = Var(kd, "float64", [1, None] if ax == 0 else [None, 1])
            shapes[kd] = tuple(kshape)
            nodes.append(Node("DimShuffle", [red], [kd], {"new_order": ["x", 0] if ax == 0 else [0, "x"]}))
            o = ew("sub" if kind == "center" else "add", [a, kd])
        elif kind == "sum":
            ax = int(rng.integers(0, 2))
            o = new((shapes[a][1 - ax],))
            nodes.append(Node("CAReduce", [a], [o], {"scalar_op": "add", "axis": [ax], "acc_dtype": "float64"}))
        elif kind == "dot":
            bs = [v for v in mats if shapes[v][0] == shapes[a][1]]
            if not bs or len([n_ for n_ in nodes if n_.op == "Dot22"]) >= 3:
                continue
            b = int(rng.choice(bs))
            o = new((shapes[a][0], shapes[b][1]))
            nodes.append(Node("Dot22", [a, b], [o], {}))
    cand = [v for v in vs if vs[v].const is None and vs[v].dtype == "float64"]
    outs = [int(v) for v in rng.choice(cand, size=int(rng.integers(1, 4)))]
    return Plan("fuzz_dag", vs, inputs, outs, nodes), shapes, idx_in


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AESARA_FUZZ_SEEDS", "8"))))
def test_random_multi_node_plans_eager_and_replay(seed):
    """Buffer planning under fire: random DAGs (``_rand_dag``) evaluated eagerly, replayed with
    fresh outputs and replayed with borrowed outputs, three calls each with NEW input tensors —
    every output of every call bit-equal to the oracle.  Catches an in-place update of a value that
    is still read, an arena range reused while a view of it is alive, a stale rebinding."""
    import interp
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(4000 + seed)
    for trial in range(25):
        # small extents by default; every fourth trial uses extents that reach the 16-byte-vector,
        # LDS-tiled and MFMA-tile code paths (values stay small integers: still exact)
        dims = [1, 2, 3, 4, 6, 7, 8, 9] if trial % 4 else [16, 24, 33, 48, 64, 66, 100, 128, 130]
        R, C = (int(v) for v in rng.choice(dims, 2, replace=False))
        plan, shapes, idx_in = _rand_dag(rng, R, C, int(rng.integers(3, 16)))
        exs = [PlanExecutor(plan), PlanExecutor(plan, use_graph=True),
               PlanExecutor(plan, use_graph=True, borrow=True)]
        for call in range(4):
            # calls 2 and 3 arrive with other extents (R != C != 5 keeps every node's operands
            # conformable): a new replay signature next to the first one
            Rc, Cc = (R, C) if call < 2 else (R + 2, C + 1 if C + 1 != R + 2 and C + 1 != 5 else C + 3)
            args = []
            for vid in plan.inputs:
                if vid == idx_in:
                    args.append(rng.integers(-Rc, Rc, 5).astype("int64"))
                else:
                    shp = tuple({R: Rc, C: Cc}[n] for n in shapes[vid])
                    args.append(rng.integers(-3, 4, shp).astype("float64"))
            try:
                want = interp.run_plan(plan, args)
            except (ValueError, IndexError):
                assert call >= 2     # extents that only matched by coincidence at R x C
                continue
            for k, ex in enumerate(exs):
                got = ex(*[torch.from_numpy(a).cuda() for a in args])
                for g, w in zip(got, want):
                    g = g.cpu().numpy()
                    assert g.shape == w.shape and np.array_equal(g, w), \
                        (seed, trial, call, ("eager", "replay", "borrow")[k], plan.pretty())


def test_half_tile_gemm_all_layouts_and_ragged_extents():
    """The 64x64-tile instantiation of the MFMA GEMM (mid-size problems) forced for every
    vector-staged problem: 4 layouts (A / B row-major or transposed views), M / N ragged around
    64 and 128, K a multiple of the slab, alpha / beta, fp32 and fp64, batched — against fp64
    NumPy at the bars of the big tile (Frobenius 2e-6 fp32, 1e-13 fp64)."""
    import torch
    from aesara_amd._lib import check, lib
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan, Var
    rng = np.random.default_rng(77)
    check(lib.ahip_set_param(b"gemm_half_max_tiles", 1 << 40))
    check(lib.ahip_set_param(b"gemm_half_min_tiles", 1))
    try:
        for dt in ("float32", "float64"):
            bk = 32 if dt == "float32" else 16
            vs = [Var(0, dt, [None, None]), Var(1, dt, []), Var(2, dt, [None, None]), Var(3, dt, [None, None]),
                  Var(4, dt, []), Var(5, dt, [None, None])]
            plan = Plan("half_gemm", {v.id: v for v in vs}, [0, 1, 2, 3, 4], [5],
                        [Node("Gemm", [0, 1, 2, 3, 4], [5], {"inplace": False})])
            ex = PlanExecutor(plan)
            for M, N, K in [(64, 64, bk), (128, 192, 4 * bk), (256, 320, 2 * bk), (100, 68, 3 * bk), (4, 260, bk),
                            (1024, 1024, 1024), (136, 72, 2112), (512, 640, 4096 + bk)]:
                for ta in (False, True):
                    for tb in (False, True):
                        x = rng.standard_normal((K, M) if ta else (M, K)).astype(dt)
                        y = rng.standard_normal((N, K) if tb else (K, N)).astype(dt)
                        z = rng.standard_normal((M, N)).astype(dt)
                        al, be = float(rng.choice([1.0, 0.8])), float(rng.choice([0.0, 0.4]))
                        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
                        (got,) = ex(torch.from_numpy(z).cuda(), np.asarray(al, dt), xd.t() if ta else xd,
                                    yd.t() if tb else yd, np.asarray(be, dt))
                        want = be * z.astype("float64") + al * ((x.T if ta else x).astype("float64")
                                                                 @ (y.T if tb else y).astype("float64"))
                        err = np.linalg.norm(got.cpu().numpy() - want) / np.linalg.norm(want)
                        assert err <= (2e-6 if dt == "float32" else 1e-13), (dt, M, N, K, ta, tb, err)
            vb = [Var(0, dt, [None] * 3), Var(1, dt, [None] * 3), Var(2, dt, [None] * 3)]
            bplan = Plan("half_bdot", {v.id: v for v in vb}, [0, 1], [2], [Node("BatchedDot", [0, 1], [2], {})])
            a3 = rng.standard_normal((3, 96, 4 * bk)).astype(dt)
            b3 = rng.standard_normal((3, 4 * bk, 160)).astype(dt)
            (got,) = PlanExecutor(bplan)(torch.from_numpy(a3).cuda(), torch.from_numpy(b3).cuda())
            want = a3.astype("float64") @ b3.astype("float64")
            err = np.linalg.norm(got.cpu().numpy() - want) / np.linalg.norm(want)
            assert err <= (2e-6 if dt == "float32" else 1e-13), (dt, "batched", err)
    finally:
        check(lib.ahip_set_param(b"gemm_half_max_tiles", 448))
        check(lib.ahip_set_param(b"gemm_half_min_tiles", 192))
