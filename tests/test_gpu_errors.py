"""Error behaviour and edge cases of the HIP path on a real device (mirrors the reference's
exception contract, SURVEY §8b: shape errors are ValueError, index errors IndexError, dtype
errors TypeError) and empty / ragged inputs."""
import numpy as np
import pytest

from golden_util import CASES, case_plan

pytestmark = pytest.mark.gpu


def _ex(name, **kw):
    from aesara_amd.executor import PlanExecutor
    return PlanExecutor(case_plan(next(c for c in CASES if c["name"] == name)), **kw)


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_elemwise_shape_mismatch_is_value_error():
    ex = _ex("ew_bcast0_float64")
    with pytest.raises(ValueError, match="Shapes on dimension 1 do not match"):
        ex(_t(np.zeros((3, 5))), _t(np.zeros((3, 4))))


def test_dtype_mismatch_is_type_error():
    ex = _ex("cfg1b_matrix_add")
    with pytest.raises(TypeError):
        ex(_t(np.zeros((3, 5), "float32")), _t(np.zeros((3, 5))))


def test_out_of_bounds_advanced_index_is_index_error():
    ex = _ex("advsub1")
    x = _t(np.arange(13 * 7, dtype="int32").reshape(13, 7))
    good = _t(np.array([0, -13, 12], dtype="int64"))
    bad = _t(np.array([0, 13, 1], dtype="int64"))
    i32 = _t(np.array([0, 1], dtype="int32"))
    v = _t(np.zeros(13))
    ex(x, good, i32, v)
    with pytest.raises(IndexError):
        ex(x, bad, i32, v)
    ex(x, good, i32, v)  # the flag is cleared after raising


def test_basic_index_out_of_bounds_is_index_error():
    ex = _ex("subtensor_basic")
    with pytest.raises(IndexError):
        ex(_t(np.zeros((9, 11), "int32")), np.int64(9))


def test_gemm_inner_dimension_mismatch():
    ex = _ex("gemm13_float32")
    with pytest.raises(ValueError):
        ex(_t(np.zeros((3, 5), "float32")), _t(np.zeros((3, 4), "float32")),
           _t(np.zeros((5, 5), "float32")))


def test_empty_reduction_of_max_raises_and_sum_is_identity():
    ex = _ex("red9_f32")  # sum over a (5, 0) matrix, axis=None
    (s,) = ex(_t(np.zeros((5, 0), "float32")))
    assert s.item() == 0.0
    exm = _ex("red0_f64")  # sum, prod, max, min of a matrix
    with pytest.raises(ValueError, match="zero-size"):
        exm(_t(np.zeros((0, 6))))


def test_ragged_sizes_elemwise_and_reduce():
    """Sizes that defeat every vector width / tile multiple (tail handling)."""
    import torch
    ex = _ex("red_fused_elemwise_axis")  # outputs: sum0, sum1, (e*2).sum(), e
    for shape in ((1, 1), (7, 13), (129, 255), (1023, 3), (1, 4097)):
        x = torch.randn(*shape, dtype=torch.float64, device="cuda")
        y = torch.randn(shape[1], dtype=torch.float64, device="cuda")
        e = torch.exp(-(x - y) ** 2)
        s0, s1, tot, em = ex(x, y)
        assert torch.allclose(em, e, rtol=1e-13, atol=0)
        assert torch.allclose(s0, e.sum(0), rtol=1e-12, atol=1e-13)
        assert torch.allclose(s1, e.sum(1), rtol=1e-12, atol=1e-13)
        assert abs(tot.item() - 2 * e.sum().item()) <= 1e-12 * abs(tot.item()) + 1e-13


def test_ragged_gemm_sizes_and_offsets():
    import torch
    ex = _ex("gemm0_float64")  # b*z + a*dot(x, y) with a=1, b=0
    for (M, K, N) in ((1, 1, 1), (5, 3, 2), (127, 129, 131), (128, 17, 256), (257, 64, 1)):
        x = torch.randn(M, K, dtype=torch.float64, device="cuda")
        y = torch.randn(K, N, dtype=torch.float64, device="cuda")
        z = torch.zeros(M, N, dtype=torch.float64, device="cuda")
        (out,) = ex(z, x, y)
        assert torch.allclose(out, x @ y, rtol=1e-12, atol=1e-12)
    # misaligned (odd element offset) views force the scalar staging path
    big = torch.randn(130, 131, dtype=torch.float64, device="cuda")
    x, y = big[1:129, 1:130], big[:129, 2:131]
    (out,) = ex(torch.zeros(128, 129, dtype=torch.float64, device="cuda"), x, y)
    assert torch.allclose(out, x @ y, rtol=1e-12, atol=1e-12)


def test_scan_zero_steps_and_one_step():
    import torch
    ex = _ex("scan_cumsum")  # outputs: res, res[-1]  -> res[-1] invalid for 0 steps, so 1 step
    x = torch.randn(1, 7, dtype=torch.float64, device="cuda")
    s0 = torch.randn(7, dtype=torch.float64, device="cuda")
    res, last = ex(x, s0)
    assert torch.allclose(res[0], s0 + x[0]) and torch.allclose(last, s0 + x[0])


def _one_node_plan(op, in_vars, out_var, params):
    from aesara_amd.plan import Node, Plan, Var
    vs = {i: Var(i, dt, list(sh)) for i, (dt, sh) in enumerate(in_vars + [out_var])}
    n = len(in_vars)
    return Plan("one", vs, list(range(n)), [n], [Node(op, list(range(n)), [n], params)])


def test_sgn_of_nan_is_nan_like_the_reference():
    """Sgn.c_code (scalar/basic.py:2620) / np.sign: NaN stays NaN."""
    import torch
    from aesara_amd.executor import PlanExecutor
    sc = {"n_in": 1, "nodes": [{"op": "sgn", "in": [["i", 0]], "dtype": "float32"}], "out": [["t", 0]]}
    ex = PlanExecutor(_one_node_plan("Elemwise", [("float32", [None])], ("float32", [None]), {"scalar": sc}))
    x = np.array([-2.5, 0.0, -0.0, 3.0, np.nan, np.inf, -np.inf], "float32")
    (got,) = ex(_t(x))
    np.testing.assert_array_equal(got.cpu().numpy(), np.sign(x))


def test_batched_dot_with_more_than_65535_items():
    """grid.z carries the batch index: a matmul over 70 000 small matrices runs as sliced
    launches instead of failing."""
    import torch
    from aesara_amd.executor import PlanExecutor
    ex = PlanExecutor(_one_node_plan("BatchedDot", [("float32", [None, None, None])] * 2,
                                     ("float32", [None, None, None]), {}))
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    A = torch.randn(70000, 4, 5, dtype=torch.float32, device="cuda", generator=g)
    B = torch.randn(70000, 5, 3, dtype=torch.float32, device="cuda", generator=g)
    (got,) = ex(A, B)
    ref = torch.einsum("bij,bjk->bik", A.double(), B.double())
    assert got.shape == (70000, 4, 3)
    assert (got.double() - ref).abs().max().item() < 1e-5


def test_uint64_index_beyond_int64_is_out_of_bounds():
    """An unsigned index >= 2**63 must not wrap into a valid negative one."""
    import torch
    from aesara_amd.executor import PlanExecutor
    ex = PlanExecutor(_one_node_plan("AdvancedSubtensor1", [("float64", [None, None]), ("uint64", [None])],
                                     ("float64", [None, None]), {}))
    x = _t(np.arange(12.0).reshape(4, 3))
    ok = torch.from_numpy(np.array([0, 3, 1], dtype=np.uint64).view(np.int64)).cuda().view(torch.uint64)
    (got,) = ex(x, ok)
    np.testing.assert_array_equal(got.cpu().numpy(), np.arange(12.0).reshape(4, 3)[[0, 3, 1]])
    bad = torch.from_numpy(np.array([0, 2 ** 64 - 1, 1], dtype=np.uint64).view(np.int64)).cuda().view(torch.uint64)
    with pytest.raises(IndexError):
        ex(x, bad)


def test_rebinding_only_touches_declared_pointer_words():
    """a scalar of a recorded argument block whose BIT PATTERN equals an
    address inside a rebindable range (here: the fill value of an int64 fill is the address of
    its own destination) must never be patched — only the words the launch site declared as
    device pointers are."""
    import ctypes as C
    import torch
    from aesara_amd._lib import check, lib
    a = torch.zeros(1024, dtype=torch.int64, device="cuda")
    b = torch.zeros(1024, dtype=torch.int64, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    addr = a.data_ptr() + 64                       # a value INSIDE the rebindable range of `a`
    val = C.c_int64(addr)
    lst = C.c_void_p()
    check(lib.ahip_list_begin())
    try:
        check(lib.ahip_fill(4, C.byref(val), C.c_void_p(a.data_ptr()), 1024, stream))
    finally:
        check(lib.ahip_list_end(C.byref(lst)))
    lo = (C.c_uint64 * 1)(a.data_ptr())
    hi = (C.c_uint64 * 1)(a.data_ptr() + 8192)
    assert lib.ahip_list_bind_bases(lst, lo, hi, 1) == 1            # the pointer, not the value
    bases = (C.c_uint64 * 1)(b.data_ptr())
    check(lib.ahip_list_run_rebased(lst, bases, 1, stream))
    torch.cuda.synchronize()
    assert int(a.abs().max()) == 0 and bool((b == addr).all())      # new buffer, ORIGINAL value
    lib.ahip_list_destroy(lst)
    # a launch recorded without a pointer map makes the list non-rebindable (never scanned)
    from aesara_amd.device import load_kernels
    (k,) = load_kernels('extern "C" __global__ void kk(long* p){ if (threadIdx.x == 0) p[0] = 7; }', ("kk",))
    arg = (C.c_void_p * 1)(a.data_ptr())
    check(lib.ahip_list_begin())
    try:
        check(lib.ahip_launch(k, 1, 1, 1, 64, 1, 1, 0, arg, 8, stream))
    finally:
        check(lib.ahip_list_end(C.byref(lst)))
    assert lib.ahip_list_bind_bases(lst, lo, hi, 1) == -2
    lib.ahip_list_destroy(lst)


def test_more_nonmergeable_dims_than_a_kernel_takes_and_2d_batched_dot():
    """two `NotImplementedError`s of the hot path are gone — an Elemwise over
    8 dims whose operands alternate broadcast / full extents (nothing merges: one launch per index
    of the outermost dims), and `BatchedDot` with 2-d operands (a batch of vectors,
    tensor/blas.py:2196-2224)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan
    p = Plan("many_dims", {}, [], [], [])
    a = p.new_var("float64", [None, 1, None, 1, None, 1, None, 1])
    b = p.new_var("float64", [1, None, 1, None, 1, None, 1, None])
    o = p.new_var("float64", [None] * 8)
    p.inputs, p.outputs = [a, b], [o]
    p.nodes = [Node("Elemwise", [a, b], [o], {"scalar": {"n_in": 2, "nodes": [
        {"op": "mul", "in": [["i", 0], ["i", 1]], "dtype": "float64"},
        {"op": "add", "in": [["t", 0], ["i", 0]], "dtype": "float64"}], "out": [["t", 1]]}})]
    rng = np.random.default_rng(0)
    av, bv = rng.standard_normal((2, 1, 3, 1, 2, 1, 3, 1)), rng.standard_normal((1, 3, 1, 2, 1, 3, 1, 2))
    for use_graph in (False, True):
        ex = PlanExecutor(p, use_graph=use_graph)
        for _ in range(2):
            (r,) = ex(_t(av), _t(bv))
            np.testing.assert_allclose(r.cpu().numpy(), av * bv + av, rtol=1e-15)
    for xs, ys in (((5, 7), (5, 7, 3)), ((5, 4, 7), (5, 7)), ((5, 7), (5, 7))):
        q = Plan("bd", {}, [], [], [])
        x = q.new_var("float32", [None] * len(xs))
        y = q.new_var("float32", [None] * len(ys))
        z = q.new_var("float32", [None] * (len(xs) + len(ys) - 3))
        q.inputs, q.outputs = [x, y], [z]
        q.nodes = [Node("BatchedDot", [x, y], [z], {})]
        xv, yv = rng.standard_normal(xs).astype("float32"), rng.standard_normal(ys).astype("float32")
        want = np.stack([np.dot(xv[i], yv[i]) for i in range(5)])
        (r,) = PlanExecutor(q)(_t(xv), _t(yv))
        assert tuple(r.shape) == want.shape
        np.testing.assert_allclose(r.cpu().numpy(), want, rtol=2e-5, atol=1e-5)
    with pytest.raises(TypeError):
        PlanExecutor(q)(_t(xv[:4]), _t(yv))


def test_float_functions_of_host_scalars_compute_in_the_output_type():
    """aesara.function([int8 scalar], op(x)) hands the executor HOST 0-d values (ScalarType inputs,
    tests/scalar/test_basic.py:319 TestUpgradeToFloat): the result of an upgrade_to_float op is
    computed in the OUTPUT type (scalar/basic.py c_code: operands convert on use) — reciprocal(int8)
    is 1 / float32(x), not NumPy's integer reciprocal or a float16 detour — and an op outside the
    host glue's integer arithmetic (arctan2) runs as a kernel on 0-d device values."""
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan

    def plan(op, n_in):
        p = Plan("host_scalar_" + op, {}, [], [], [])
        ins = [p.new_var("int8", []) for _ in range(n_in)]
        o = p.new_var("float32", [])
        p.inputs, p.outputs = ins, [o]
        p.nodes = [Node("Elemwise", ins, [o], {"scalar": {"n_in": n_in, "nodes": [
            {"op": op, "in": [["i", k] for k in range(n_in)], "dtype": "float32"}], "out": [["t", 0]]}})]
        return p
    for op, ref in (("reciprocal", lambda v: np.float32(1) / v), ("exp", np.exp), ("sqrt", np.sqrt),
                    ("log", np.log), ("cos", np.cos), ("deg2rad", np.deg2rad)):
        ex = PlanExecutor(plan(op, 1))
        for x in (-127, -3, 2, 5, 88):
            if op in ("sqrt", "log") and x < 0:
                continue
            (r,) = ex(np.int8(x))
            r = np.asarray(r.cpu() if hasattr(r, "cpu") else r)
            assert r.dtype == np.float32 and r.shape == ()
            np.testing.assert_allclose(r, ref(np.float32(x)), rtol=2e-6)
    ex = PlanExecutor(plan("arctan2", 2))
    for x, y in ((-127, 3), (5, -7), (0, 0), (88, 127)):
        (r,) = ex(np.int8(x), np.int8(y))
        r = np.asarray(r.cpu() if hasattr(r, "cpu") else r)
        np.testing.assert_allclose(r, np.arctan2(np.float32(x), np.float32(y)), rtol=2e-6, atol=1e-7)
