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
