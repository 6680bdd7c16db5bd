"""Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle cannot
run these shapes in seconds), plus torch-fp64 spot restatements on sub-blocks.  All through the
HIP path (plans lowered from the reference graphs; kernels via the C-ABI)."""
import numpy as np
import pytest

from golden_util import CASES, case_plan

pytestmark = pytest.mark.gpu


def _plan(name):
    return case_plan(next(c for c in CASES if c["name"] == name))


def _ex(name, **kw):
    from aesara_amd.executor import PlanExecutor
    return PlanExecutor(_plan(name), **kw)


def _randn(shape, dtype, seed, scale=1.0):
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randn(*shape, dtype=dtype, device="cuda", generator=g) * scale


def test_cfg2_full_size_block_additivity_and_fp64_restatement():
    """exp(-(x-mu)^2/2s^2).sum() on fp64 4096x4096: the sum over the whole matrix equals the sum
    of the sums over its four row blocks (tree order differs: 1e-12), equals a torch fp64
    restatement (north_star tolerance 1e-6 rel; we hold 1e-12), and is deterministic."""
    import torch
    ex = _ex("cfg2_gauss_sum")
    x = _randn((4096, 4096), torch.float64, 1)
    mu = torch.tensor(0.1, dtype=torch.float64, device="cuda")
    sg = torch.tensor(1.3, dtype=torch.float64, device="cuda")
    full = ex(x, mu, sg)[0].item()
    parts = sum(ex(x[i * 1024:(i + 1) * 1024], mu, sg)[0].item() for i in range(4))
    want = torch.exp(-(x - 0.1) ** 2 / (2 * 1.3 ** 2)).sum().item()
    assert abs(full - parts) <= 1e-12 * abs(full)
    assert abs(full - want) <= 1e-12 * abs(want)
    assert ex(x, mu, sg)[0].item() == full  # bitwise run-to-run determinism
    # replay path gives the same bits as the eager path
    exg = _ex("cfg2_gauss_sum", use_graph=True)
    for _ in range(3):
        got = exg(x, mu, sg)[0].item()
    assert got == full


def test_cfg1b_full_size_add_is_bit_exact():
    import torch
    ex = _ex("cfg1b_matrix_add")
    x, y = _randn((4096, 4096), torch.float64, 0), _randn((4096, 4096), torch.float64, 1)
    (z,) = ex(x, y)
    assert torch.equal(z, x + y)
    (zt,) = ex(x, y.t())  # strided operand path
    assert torch.equal(zt, x + y.t())


def test_cfg3a_gemv_full_size_linearity_and_rows():
    import torch
    ex = _ex("gemv_beta_float64")  # beta*y + alpha*dot(A, x), alpha=0.5 beta=-0.3
    M = _randn((4096, 4096), torch.float64, 2)
    v1, v2 = _randn((4096,), torch.float64, 3), _randn((4096,), torch.float64, 4)
    y0 = torch.zeros(4096, dtype=torch.float64, device="cuda")
    a = ex(y0, M, v1)[0]
    b = ex(y0, M, v2)[0]
    ab = ex(y0, M, v1 + v2)[0]
    assert torch.allclose(ab, a + b, rtol=1e-11, atol=1e-9)            # linearity in x
    assert torch.allclose(a, 0.5 * (M @ v1), rtol=1e-10, atol=1e-9)     # north-star 1e-10 (fp64)
    at = ex(y0, M.t(), v1)[0]                                           # column layout
    assert torch.allclose(at, 0.5 * (M.t() @ v1), rtol=1e-10, atol=1e-9)


def test_cfg3b_gemm_full_size_frobenius_and_associativity():
    """Gemm 4096^3 fp32 (0.4*C + 0.8*A@B): ||C - C_ref||_F / ||C_ref||_F <= 1e-6 against an
    fp64 restatement (SURVEY §8d norm), on all four operand layouts; (A@B)@v == A@(B@v)."""
    import torch
    ex = _ex("cfg3b_gemm_update")
    Cm = _randn((4096, 4096), torch.float32, 1)
    A, B = _randn((4096, 4096), torch.float32, 3), _randn((4096, 4096), torch.float32, 4)
    ref = (0.4 * Cm.double() + 0.8 * (A.double() @ B.double()))
    for a, b, refm in ((A, B, ref),
                       (A.t(), B, 0.4 * Cm.double() + 0.8 * (A.double().t() @ B.double())),
                       (A, B.t(), 0.4 * Cm.double() + 0.8 * (A.double() @ B.double().t())),
                       (A.t(), B.t(), 0.4 * Cm.double() + 0.8 * (A.double().t() @ B.double().t()))):
        (out,) = ex(Cm, a, b)
        rel = torch.linalg.norm(out.double() - refm) / torch.linalg.norm(refm)
        assert rel.item() <= 1e-6, rel.item()
    v = _randn((4096,), torch.float32, 9).double()
    (out,) = ex(torch.zeros_like(Cm), A, B)
    lhs = out.double() @ v
    rhs = 0.8 * (A.double() @ (B.double() @ v))
    assert (torch.linalg.norm(lhs - rhs) / torch.linalg.norm(rhs)).item() <= 1e-5


@pytest.mark.parametrize("M,N,K", [(64, 1024, 1024), (8, 1024, 4096), (50, 1000, 1004),
                                   (37, 333, 1003), (1024, 1024, 1024), (64, 256, 20480),
                                   (1024, 1000, 8192), (2100, 2500, 512)])
def test_small_output_gemm_kernels_all_layouts(M, N, K):
    """(The last shape is a large ragged problem: aligned block unpredicated + edge strips.)
    Outputs too small for one 128x128 tile per CU take the 16-row split-K kernels (vector-load
    form for k-/n-contiguous operands, scalar-load form otherwise): Frobenius error <= 1e-6
    (fp32) / 1e-13 (fp64) against an fp64 restatement, all four operand layouts, beta != 0."""
    import torch
    ex = _ex("cfg3b_gemm_update")               # fp32: 0.4*C + 0.8*A@B
    for dt, e, tol in ((torch.float32, ex, 1e-6), (torch.float64, None, 1e-13)):
        Cm = _randn((M, N), dt, 1)
        A, At = _randn((M, K), dt, 2), _randn((K, M), dt, 3).t()
        B, Bt = _randn((K, N), dt, 4), _randn((N, K), dt, 5).t()
        for a in (A, At):
            for b in (B, Bt):
                if dt == torch.float32:
                    (out,) = e(Cm, a, b)
                    ref = 0.4 * Cm.double() + 0.8 * (a.double() @ b.double())
                else:
                    out = _gemm64(Cm, a, b)
                    ref = Cm + a @ b
                rel = torch.linalg.norm(out.double() - ref) / torch.linalg.norm(ref)
                assert rel.item() <= tol, (M, N, K, str(dt), rel.item())


def _gemm64(Cm, a, b):
    """C + A@B in fp64 through ahip_gemm (executor's Gemm path on a one-node plan)."""
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan
    global _GEMM64
    try:
        ex = _GEMM64
    except NameError:
        p = Plan("gemm64", {}, [], [], [])
        z, x, y = (p.new_var("float64", [None, None], n) for n in "zxy")
        one = p.add_const(np.float64(1.0), dtype="float64")
        o = p.new_var("float64", [None, None], "o")
        p.inputs = [z, x, y]
        p.nodes.append(Node("Gemm", [z, one, x, y, one], [o], {"inplace": False}))
        p.outputs = [o]
        ex = _GEMM64 = PlanExecutor(p)
    return ex(Cm, a, b)[0]


def test_cfg4_scan_full_size_chunk_consistency():
    """GRU scan T=512, H=1024 fp32: running steps [0,256) then [256,512) from the carried state
    equals one 512-step scan (recurrence semantics), within the fp32 recurrence tolerance, and the
    hipGraph replay equals the eager run bit-for-bit."""
    import torch
    T, H = 512, 1024
    x = _randn((T, H), torch.float32, 4, 0.1)
    h0 = torch.zeros(H, dtype=torch.float32, device="cuda")
    Ws = [_randn((H, H), torch.float32, 5 + k, 1.0 / np.sqrt(H)) for k in range(6)]
    ex = _ex("cfg4_gru_b1_f32")
    hs, hT = ex(x, h0, *Ws)
    hs = hs.clone(); hT = hT.clone()
    assert hs.shape == (T, H) and torch.isfinite(hs).all()
    assert torch.equal(hs[-1], hT)
    hs_a, h_mid = ex(x[:256], h0, *Ws)
    h_mid = h_mid.clone()
    hs_b, h_end = ex(x[256:], h_mid, *Ws)
    assert torch.allclose(hs[:256], hs_a, rtol=0, atol=0)
    assert torch.allclose(hs[256:], hs_b, rtol=1e-5, atol=1e-6)
    # fp64 restatement of the last 8 steps from the device state at T-8
    def step(xt, h):
        z = torch.sigmoid(xt @ Ws[0].double() + h @ Ws[1].double())
        r = torch.sigmoid(xt @ Ws[2].double() + h @ Ws[3].double())
        hh = torch.tanh(xt @ Ws[4].double() + (r * h) @ Ws[5].double())
        return (1 - z) * h + z * hh
    h = hs[T - 9].double()
    for t in range(T - 8, T):
        h = step(x[t].double(), h)
    assert torch.allclose(hT.double(), h, rtol=1e-5, atol=1e-6)
    exg = _ex("cfg4_gru_b1_f32", use_graph=True)
    for _ in range(2):
        hs_g, _ = exg(x, h0, *Ws)
    assert torch.equal(hs_g, hs)


def test_cfg5_logistic_large_batch_matches_fp64_restatement():
    """logp + grad at N = 2^20 x 256 fp32 (1 GiB of X) against a torch fp64 restatement:
    logp, db rel <= 1e-6 (fp64 accumulators, CAReduce._acc_dtype), dw <= 1e-5."""
    import torch
    N, D = 1 << 20, 256
    X = _randn((N, D), torch.float32, 6)
    w = _randn((D,), torch.float32, 7, 1.0 / 16)
    b = torch.tensor(0.1, dtype=torch.float32, device="cuda")
    y = (torch.rand(N, device="cuda") < 0.5).float()
    logp, gw, gb = _ex("cfg5_logistic")(X, w, b, y)
    z = X.double() @ w.double() + 0.1
    yd = y.double()
    ref_logp = -(yd * torch.nn.functional.softplus(-z) + (1 - yd) * torch.nn.functional.softplus(z)).sum()
    r = yd - torch.sigmoid(z)
    assert abs(logp.item() - ref_logp.item()) <= 2e-6 * abs(ref_logp.item())
    assert abs(gb.item() - r.sum().item()) <= 1e-5 * max(1.0, abs(r.sum().item())) + 1e-2
    ref_gw = X.double().t() @ r
    assert (torch.linalg.norm(gw.double() - ref_gw) / torch.linalg.norm(ref_gw)).item() <= 1e-5
    # row-block additivity of every output (the property the multi-GPU sharding relies on)
    halves = [_ex("cfg5_logistic")(X[i * N // 2:(i + 1) * N // 2], w, b, y[i * N // 2:(i + 1) * N // 2])
              for i in range(2)]
    assert abs(sum(h[0].item() for h in halves) - logp.item()) <= 2e-6 * abs(logp.item())
    assert torch.allclose(halves[0][1] + halves[1][1], gw, rtol=1e-4, atol=1e-2)


def test_softmax_rowchain_large_rows_match_fp64_restatement():
    """Row-chain kernel at 32768 x 1000 fp32 (ragged K: not a multiple of the 256-element chunk)
    and 4096 x 4096: rows sum to 1, equal a torch fp64 softmax to fp32 round-off, and the
    unfused three-pass path gives the same values."""
    import torch
    for shape in ((32768, 1000), (4096, 4096), (100000, 10)):
        x = _randn(shape, torch.float32, 3, 4.0)
        (y,) = _ex("softmax_rows_f32")(x)
        ref = torch.softmax(x.double(), dim=-1)
        assert y.shape == x.shape and torch.isfinite(y).all()
        assert (y.double().sum(-1) - 1).abs().max().item() <= 2e-6
        assert torch.allclose(y.double(), ref, rtol=3e-6, atol=1e-9)
        (yu,) = _ex("softmax_rows_f32", fuse=False)(x)
        assert torch.allclose(y, yu, rtol=2e-6, atol=1e-9)
    # extreme logits: the max-shift keeps everything finite (tests/tensor/test_special.py)
    x = torch.tensor([[1e4, 0.0, -1e4], [88.0, 89.0, 90.0]], dtype=torch.float32, device="cuda")
    (y,) = _ex("softmax_rows_f32")(x)
    assert torch.allclose(y.double(), torch.softmax(x.double(), -1), rtol=1e-6, atol=1e-12)


def test_long_row_chains_vocabulary_sized():
    """Rows that do not fit a wavefront's registers take the one-workgroup-per-row form of the
    row-chain kernel (every reduction a sweep, intermediates recomputed): softmax over a 50 304
    wide vocabulary, log-softmax fp64, layer norm over 32 768 columns; against torch fp64 and
    against the unfused node-by-node path."""
    import torch
    x = _randn((2048, 50304), torch.float32, 3, 4.0)
    (y,) = _ex("softmax_rows_f32")(x)
    ref = torch.softmax(x.double(), dim=-1)
    assert (y.double().sum(-1) - 1).abs().max().item() <= 5e-6
    assert torch.allclose(y.double(), ref, rtol=3e-6, atol=1e-10)
    (yu,) = _ex("softmax_rows_f32", fuse=False)(x)
    assert torch.allclose(y, yu, rtol=2e-6, atol=1e-10)
    xd = _randn((300, 20002), torch.float64, 4, 3.0)      # K % 4 != 0, K % 2 == 0
    (ld,) = _ex("logsoftmax_rows_f64")(xd)
    assert torch.allclose(ld, torch.log_softmax(xd, dim=-1), rtol=1e-12, atol=1e-12)
    xl = _randn((8, 16, 32768), torch.float32, 5, 2.0) + 0.5
    g, b = _randn((32768,), torch.float32, 6), _randn((32768,), torch.float32, 7)
    outs = _ex("layernorm_float32")(xl, g, b)
    outs_u = _ex("layernorm_float32", fuse=False)(xl, g, b)
    for u, v in zip(outs, outs_u):
        assert torch.allclose(u.double(), v.double(), rtol=3e-5, atol=3e-5)


def test_layernorm_rowchain_full_size():
    import torch
    x = _randn((64, 512, 1024), torch.float32, 5, 2.0) + 0.5
    g, b = _randn((1024,), torch.float32, 6), _randn((1024,), torch.float32, 7)
    y, mu, var = _ex("layernorm_float32")(x, g, b)
    xd = x.double()
    mud = xd.mean(-1, keepdim=True)
    vard = ((xd - mud) ** 2).mean(-1, keepdim=True)
    ref = (xd - mud) / torch.sqrt(vard + 1e-5) * g.double() + b.double()
    assert torch.allclose(mu.double(), mud, rtol=1e-5, atol=1e-6)
    assert torch.allclose(var.double(), vard[..., 0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(y.double(), ref, rtol=2e-5, atol=2e-5)


def test_cfg5_at_baseline_full_size_16gib():
    """BASELINE config 5 at its full shape: N = 2^24 rows x 256 fp32 (16 GiB of X, 4.3e9 elements:
    64-bit indexing throughout), one pass of the row-program kernel; logp / d/dw against an fp64
    restatement accumulated over row blocks (logp rel <= 1e-6, dw <= 1e-5)."""
    import torch
    free, _total = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs ~40 GiB of free HBM")
    N, D = 1 << 24, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(6)
    X = torch.empty((N, D), dtype=torch.float32, device="cuda")
    for i in range(0, N, 1 << 20):
        X[i:i + (1 << 20)] = torch.randn((1 << 20, D), dtype=torch.float32, device="cuda", generator=g)
    w = torch.randn((D,), dtype=torch.float32, device="cuda", generator=g) / 16
    b = torch.tensor(0.1, dtype=torch.float32, device="cuda")
    y = (torch.rand(N, device="cuda", generator=g) < 0.5).float()
    logp, gw, gb = _ex("cfg5_logistic")(X, w, b, y)
    ref_logp, ref_gb = 0.0, 0.0
    ref_gw = torch.zeros(D, dtype=torch.float64, device="cuda")
    for i in range(0, N, 1 << 21):
        Xd, yd = X[i:i + (1 << 21)].double(), y[i:i + (1 << 21)].double()
        z = Xd @ w.double() + 0.1
        ref_logp += -(yd * torch.nn.functional.softplus(-z)
                      + (1 - yd) * torch.nn.functional.softplus(z)).sum().item()
        r = yd - torch.sigmoid(z)
        ref_gb += r.sum().item()
        ref_gw += Xd.t() @ r
    assert abs(logp.item() - ref_logp) <= 1e-6 * abs(ref_logp)
    assert (torch.linalg.norm(gw.double() - ref_gw) / torch.linalg.norm(ref_gw)).item() <= 1e-5
    assert abs(gb.item() - ref_gb) <= 1e-5 * abs(ref_gb) + 0.5


def test_large_index_ops_bit_exact():
    """Row gather / scatter-add on 2^20 indices: bit-exact against torch integer ops."""
    import torch
    n_rows, n_idx = 50_000, 1 << 20
    x = torch.randint(-1000, 1000, (n_rows, 7), dtype=torch.int32, device="cuda")
    idx = torch.randint(-n_rows, n_rows, (n_idx,), dtype=torch.int64, device="cuda")
    ex = _ex("advsub1")  # inputs: x (imatrix), idx (int64), idx32, v
    idx32 = torch.randint(-7, 7, (5,), dtype=torch.int32, device="cuda")
    v = _randn((n_rows,), torch.float64, 1)
    outs = ex(x, idx, idx32, v)
    wrapped = torch.where(idx < 0, idx + n_rows, idx)
    assert torch.equal(outs[0], x[wrapped])
    assert torch.equal(outs[2], v[wrapped])
    y = torch.randint(-5, 5, (n_idx, 7), dtype=torch.int32, device="cuda")
    ex2 = _ex("advincsub1_int")
    got = ex2(x, y, idx)[0]
    want = x.clone().index_add_(0, wrapped, y)
    assert torch.equal(got, want)


def test_fused_kernels_agree_with_unfused_on_random_shapes():
    """Shape fuzzing of the generated fused kernels (row chains, GEMM-chain epilogues, GEMV chains,
    small-output GEMMs) against the unfused node-by-node path on ragged / degenerate shapes:
    odd K, K smaller than a vector, single rows, sizes around the tile edges."""
    import torch
    rng = np.random.default_rng(123)

    def close(a, b, tol):
        return torch.allclose(a.double(), b.double(), rtol=tol, atol=tol)

    sizes = [1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, 255, 257, 1000]
    # softmax / log-softmax / layer norm: rows x K
    for _ in range(25):
        n, k = int(rng.choice(sizes)), int(rng.choice(sizes[1:]))
        x = _randn((n, k), torch.float32, int(rng.integers(1 << 30)), 3.0)
        (a,) = _ex("softmax_rows_f32")(x)
        (b,) = _ex("softmax_rows_f32", fuse=False)(x)
        assert close(a, b, 2e-6), (n, k)
        xd = x.double()
        (a,) = _ex("logsoftmax_rows_f64")(xd)
        (b,) = _ex("logsoftmax_rows_f64", fuse=False)(xd)
        assert close(a, b, 1e-12), (n, k)
    for _ in range(10):
        a_, b_, k = (int(rng.choice(sizes[:10])) for _ in range(3))
        k = max(k, 2)
        x = _randn((a_, b_, k), torch.float32, int(rng.integers(1 << 30)), 2.0)
        g, bb = _randn((k,), torch.float32, 1), _randn((k,), torch.float32, 2)
        for u, v in zip(_ex("layernorm_float32")(x, g, bb), _ex("layernorm_float32", fuse=False)(x, g, bb)):
            assert close(u, v, 3e-5), (a_, b_, k)
    # small-batch layers: x[m,k1] y[m,k2] W,W2[k1,n] U[k2,n] Wt[n,k1] b[n]
    for _ in range(20):
        m, k1, k2, n = (int(rng.choice(sizes[:16])) for _ in range(4))
        s = int(rng.integers(1 << 30))
        args = [_randn((m, k1), torch.float32, s), _randn((m, k2), torch.float32, s + 1),
                _randn((k1, n), torch.float32, s + 2, 0.3), _randn((k1, n), torch.float32, s + 3, 0.3),
                _randn((k2, n), torch.float32, s + 4, 0.3), _randn((n, k1), torch.float32, s + 5, 0.3),
                _randn((n,), torch.float32, s + 6)]
        for u, v in zip(_ex("mlp_layers_float32")(*args), _ex("mlp_layers_float32", fuse=False)(*args)):
            assert close(u, v, 2e-5), (m, k1, k2, n)


def _reduce_plan(dt, nd, axis, op, out_dt=None, acc=None):
    from aesara_amd.plan import Node, Plan, Var
    out_dt = out_dt or dt
    return Plan("axisred", {0: Var(0, dt, [None] * nd), 1: Var(1, out_dt, [None] * (nd - len(axis)))},
                [0], [1], [Node("CAReduce", [0], [1], {"scalar_op": op, "axis": list(axis),
                                                       "acc_dtype": acc or out_dt})])


def test_axis_reductions_all_layouts_against_torch():
    """Axis CAReduce over shapes that exercise every layout of the generated kernels: short and
    long rows (1-64 lanes per output), vector and scalar loads (extents not divisible by the
    vector width, transposed views), sliced long reductions in both the row and the column
    form, 3-d mixed axes.  Integer / bool results bit-exact, float sums vs fp64."""
    import torch
    from aesara_amd.device import DevArray
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(7)
    shapes = [((1 << 18, 8), (0,)), ((1 << 18, 8), (1,)), ((8, 1 << 18), (1,)), ((8, 1 << 18), (0,)),
              ((5, 100003), (1,)), ((100003, 5), (0,)), ((3001, 257), (0,)), ((3001, 257), (1,)),
              ((64, 96, 80), (0, 2)), ((64, 96, 80), (1,)), ((64, 96, 80), (0, 1)), ((7, 3, 50001), (2,)),
              ((1, 4096), (1,)), ((4096, 1), (0,)), ((2, 2), (0,)), ((70000, 3), (1,))]
    for shape, axis in shapes:
        n = int(np.prod(shape))
        # float sums (fp32 accumulates in fp64: tensor/elemwise.py:1371 _acc_dtype), max with NaN
        for dt, tdt, acc, tol in (("float64", torch.float64, "float64", 1e-12),
                                  ("float32", torch.float32, "float64", 2e-6)):
            x = _randn(shape, tdt, int(rng.integers(1 << 30)))
            (got,) = PlanExecutor(_reduce_plan(dt, len(shape), axis, "add", acc=acc))(x)
            want = x.double().sum(dim=axis).to(tdt)
            assert torch.allclose(got, want, rtol=tol, atol=tol * max(1.0, n ** 0.5 / 8)), (shape, axis, dt)
            xn = x.clone()
            xn.view(-1)[int(rng.integers(n))] = float("nan")
            (got,) = PlanExecutor(_reduce_plan(dt, len(shape), axis, "maximum"))(xn)
            want = xn
            for a in sorted(axis, reverse=True):
                want = want.max(dim=a).values    # torch.max propagates NaN like the reference
            assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0)), \
                (shape, axis, dt)
        xi = torch.randint(-1000, 1000, shape, dtype=torch.int64, device="cuda")
        (got,) = PlanExecutor(_reduce_plan("int64", len(shape), axis, "add"))(xi)
        assert torch.equal(got, xi.sum(dim=axis)), (shape, axis)
        xb = torch.rand(shape, device="cuda") < 1e-4
        (got,) = PlanExecutor(_reduce_plan("bool", len(shape), axis, "or"))(xb)
        want = xb
        for a in sorted(axis, reverse=True):
            want = want.any(dim=a)
        assert torch.equal(got, want), (shape, axis)
    # a transposed view: the unit stride moves to the other group
    x = _randn((3000, 500), torch.float32, 5)
    v = DevArray.from_torch(x).view([500, 3000], [1, 500])
    for axis in ((0,), (1,)):
        (got,) = PlanExecutor(_reduce_plan("float32", 2, axis, "add", acc="float64"))(v)
        want = x.t().double().sum(dim=axis).float()
        assert torch.allclose(got, want, rtol=2e-6, atol=2e-5)


def test_tiled_transposed_elemwise_random_shapes():
    """The LDS-tiled Elemwise kernel (transposed operands) on ragged shapes around the tile edges,
    against torch: x + y.T, exp(y.T/4)*x - v, sqr(y.T); (x*y.T).sum() & co."""
    import torch
    rng = np.random.default_rng(11)
    sizes = [16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 100, 127, 129, 200, 1000]
    for _ in range(24):
        r, c = int(rng.choice(sizes)), int(rng.choice(sizes))
        s = int(rng.integers(1 << 30))
        x, y, v = _randn((r, c), torch.float32, s), _randn((c, r), torch.float32, s + 1), \
            _randn((c,), torch.float32, s + 2)
        a, b, d = _ex("ew_transposed_float32_64x128")(x, y, v)
        assert torch.allclose(a, x + y.t(), rtol=1e-6, atol=1e-6), (r, c)
        assert torch.allclose(b, torch.exp(y.t() * 0.25) * x - v, rtol=2e-6, atol=2e-6), (r, c)
        assert torch.allclose(d, y.t() ** 2, rtol=1e-6, atol=1e-6), (r, c)
        xd, yd = x.double(), y.double()
        s0, m0, s1 = _ex("reduce_all_transposed_float64")(xd, yd)
        assert torch.allclose(s0, (xd * yd.t()).sum(), rtol=1e-11, atol=1e-9), (r, c)
        assert torch.equal(m0, (yd.t() - xd).max()), (r, c)
        assert torch.allclose(s1, (xd.t() ** 2 + yd).sum(), rtol=1e-11, atol=1e-9), (r, c)
    # the same kernels on a big ragged 3-d problem (transpose of the two inner dims)
    x, y, z, w = _randn((3, 1000, 777), torch.float64, 1), _randn((3, 777, 1000), torch.float64, 2), \
        _randn((777, 3, 1000), torch.float64, 3), _randn((777, 1000), torch.float64, 4)
    (o,) = _ex("ew_transposed_3d")(x, y, z, w)
    want = x * y.permute(0, 2, 1) + z.permute(1, 2, 0) - w.t()[None]
    assert torch.allclose(o, want, rtol=1e-13, atol=1e-13)     # mul+add contracts to an fma


def _one_node_plan(op, in_vars, out_var, params):
    from aesara_amd.plan import Node, Plan, Var
    vs = {i: Var(i, dt, [None] * nd) for i, (dt, nd) in enumerate(in_vars + [out_var])}
    n = len(in_vars)
    return Plan("one", vs, list(range(n)), [n], [Node(op, list(range(n)), [n], params)])


def test_cumulative_chunked_scans_against_torch():
    """CumOp on few long lines (flat vectors, axis 0 of matrices): the chunked reduce-then-scan
    form.  Integers bit-exact (wrapping), floats against an fp64 scan."""
    import torch
    from aesara_amd.executor import PlanExecutor
    cases = [((1 << 22) + 3,), (3000, 700), (700, 3000), (9, 40000, 5), (4, 100000), (70000, 4), (1,), (5, 1)]
    for shape in cases:
        nd = len(shape)
        for axis in range(nd):
            xi = torch.randint(-3, 4, shape, dtype=torch.int64, device="cuda")
            (got,) = PlanExecutor(_one_node_plan("CumOp", [("int64", nd)], ("int64", nd),
                                                 {"axis": axis, "mode": "add"}))(xi)
            assert torch.equal(got, xi.cumsum(dim=axis)), (shape, axis)
            x = _randn(shape, torch.float32, 17 + axis)
            (got,) = PlanExecutor(_one_node_plan("CumOp", [("float32", nd)], ("float32", nd),
                                                 {"axis": axis, "mode": "add"}))(x)
            want = x.double().cumsum(dim=axis)
            scale = float(want.abs().max()) + 1.0
            assert float((got.double() - want).abs().max()) <= 4e-6 * scale, (shape, axis)
        # products of +-1 / 2 / 1: exact in int64 for short runs, wrapping for long ones
        xm = torch.where(torch.rand(shape, device="cuda") < 0.001, -1, 1).to(torch.int64)
        (got,) = PlanExecutor(_one_node_plan("CumOp", [("int64", nd)], ("int64", nd),
                                             {"axis": 0, "mode": "mul"}))(xm)
        assert torch.equal(got, xm.cumprod(dim=0)), shape


def test_argmax_row_and_column_forms_against_torch():
    """Argmax over the last axis (wave per row) and over axis 0 (column form, sliced): first
    maximum on ties, NaN counts as the maximum (np.argmax), vector and scalar load paths."""
    import torch
    from aesara_amd.executor import PlanExecutor

    def np_argmax(x, axis):
        # np.argmax semantics with torch: NaN is the maximum, first occurrence wins
        key = torch.where(torch.isnan(x), torch.full_like(x, float("inf")), x) if x.is_floating_point() else x
        mx = key.max(dim=axis, keepdim=True).values
        hit = key == mx
        if x.is_floating_point():
            anynan = torch.isnan(x).any(dim=axis, keepdim=True)
            hit = torch.where(anynan, torch.isnan(x), hit)
        idx = torch.arange(x.shape[axis], device=x.device).reshape([-1 if d == axis else 1 for d in range(x.ndim)])
        big = x.shape[axis]
        return torch.where(hit, idx, big).min(dim=axis).values

    for shape in ((5000, 300), (300, 5000), (4099, 257), (64, 64), (100000, 16), (16, 100000)):
        for axis in (0, 1):
            xi = torch.randint(-5, 6, shape, dtype=torch.int32, device="cuda")      # many ties
            (got,) = PlanExecutor(_one_node_plan("Argmax", [("int32", 2)], ("int64", 1), {"axis": [axis]}))(xi)
            assert torch.equal(got, np_argmax(xi, axis)), (shape, axis)
            x = torch.round(_randn(shape, torch.float64, 3) * 2)
            x.view(-1)[::7919] = float("nan")
            (got,) = PlanExecutor(_one_node_plan("Argmax", [("float64", 2)], ("int64", 1), {"axis": [axis]}))(x)
            assert torch.equal(got, np_argmax(x, axis)), (shape, axis)
            xf = _randn(shape, torch.float32, 4)
            (got,) = PlanExecutor(_one_node_plan("Argmax", [("float32", 2)], ("int64", 1), {"axis": [axis]}))(xf)
            assert torch.equal(got, xf.argmax(dim=axis)), (shape, axis)


def test_sort_and_argsort_rows_against_torch_stable_sort():
    """Sort / ArgSort at the LDS capacity, odd lengths, many ties (stable order), NaNs (last),
    several dtypes and axes; rows that do not fit are refused loudly."""
    import torch
    from aesara_amd.executor import PlanExecutor

    def plans(dt, nd):
        from aesara_amd.plan import Node, Plan, Var
        out = []
        for op, odt in (("Sort", dt), ("ArgSort", "int64")):
            vs = {0: Var(0, dt, [None] * nd), 1: Var(1, "int64", []), 2: Var(2, odt, [None] * nd)}
            out.append(Plan(op, vs, [0, 1], [2], [Node(op, [0, 1], [2], {"kind": "quicksort"})]))
        return out
    for shape, axis in (((300, 4096), 1), ((4096, 70), 0), ((5, 1000, 7), 1), ((2000, 3), 1), ((1, 1), 0),
                        ((33, 1025), -1)):
        for tdt, dt in ((torch.float64, "float64"), (torch.float32, "float32"), (torch.int32, "int32"),
                        (torch.int8, "int8")):
            if tdt.is_floating_point:
                x = torch.round(_randn(shape, tdt, 9) * 3)          # many ties
                x.view(-1)[::997] = float("nan")
            else:
                x = torch.randint(-20, 20, shape, dtype=tdt, device="cuda")
            ps, pa = plans(dt, len(shape))
            (vals,) = PlanExecutor(ps)(x, np.int64(axis))
            (idx,) = PlanExecutor(pa)(x, np.int64(axis))
            wv, wi = torch.sort(x, dim=axis, stable=True)
            assert torch.equal(torch.nan_to_num(vals, nan=1e30), torch.nan_to_num(wv, nan=1e30)), (shape, axis, dt)
            assert torch.equal(idx, wi), (shape, axis, dt)
    # rows longer than LDS (round 6: chunk sort + rank-based merge passes), ragged, with NaNs and ties
    for shape, axis, dt, tdt in (((2, 5000), 1, "float64", torch.float64), ((3, 100003), 1, "float32", torch.float32),
                                 ((70001,), 0, "int32", torch.int32), ((1 << 20,), 0, "float32", torch.float32)):
        if dt.startswith("float"):
            x = torch.round(_randn(shape, tdt, 7) * 50)            # many ties
            x.view(-1)[::997] = float("nan")
        else:
            x = torch.randint(-2000, 2000, shape, dtype=tdt, device="cuda")
        ps, pa = plans(dt, len(shape))
        (vals,) = PlanExecutor(ps)(x, np.int64(axis))
        (idx,) = PlanExecutor(pa)(x, np.int64(axis))
        wv, wi = torch.sort(x, dim=axis, stable=True)
        assert torch.equal(torch.nan_to_num(vals, nan=1e30), torch.nan_to_num(wv, nan=1e30)), (shape, axis, dt)
        assert torch.equal(idx, wi), (shape, axis, dt)


@pytest.mark.parametrize("T,H,B,name", [(64, 256, 1, "gru_bptt_b1_f32"), (24, 128, 16, "gru_bptt_b4_f64"),
                                         (40, 256, 48, "gru_bptt_b4_f32"), (9, 64, 5, "gru_bptt_b4_f32"),
                                         # more 16 x 16 tiles than CUs: two (or four) batch blocks per workgroup
                                         (12, 1024, 128, "gru_bptt_b4_f32"), (10, 512, 256, "gru_bptt_b4_f32"),
                                         (6, 1024, 256, "gru_bptt_b4_f32")])      # four blocks
def test_gru_bptt_against_torch_autograd(T, H, B, name):
    """SURVEY §8(f3) at a real shape: loss and gradients of the GRU recurrence (forward Scan +
    gradient Scan with mit-mot accumulators, lowered from aesara.grad) against torch.autograd of an
    fp64 restatement — eager and replayed."""
    import torch
    from aesara_amd.executor import PlanExecutor
    dt = torch.float32 if name.endswith("f32") else torch.float64
    g = torch.Generator(device="cuda")
    g.manual_seed(13)
    shp = (T, H) if B == 1 else (T, B, H)
    x = torch.randn(*shp, dtype=dt, device="cuda", generator=g) * 0.3
    h0 = torch.randn(*shp[1:], dtype=dt, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=dt, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    leaves = [w.double().requires_grad_(True) for w in Ws] + [h0.double().requires_grad_(True)]
    Wz, Uz, Wr, Ur, Wh, Uh, h = leaves
    hs = []
    for t in range(T):
        xt = x[t].double()
        z = torch.sigmoid(xt @ Wz + h @ Uz)
        r = torch.sigmoid(xt @ Wr + h @ Ur)
        hh = torch.tanh(xt @ Wh + (r * h) @ Uh)
        h = (1 - z) * h + z * hh
        hs.append(h)
    loss = (hs[-1] ** 2).sum() + torch.stack(hs).mean()
    grads = torch.autograd.grad(loss, leaves)
    want = [loss.detach()] + [gr.detach() for gr in grads]
    tol = 2e-4 if dt == torch.float32 else 1e-9
    for use_graph in (False, True):
        ex = PlanExecutor(_plan(name), use_graph=use_graph)
        for _ in range(2):
            got = ex(x, h0, *Ws)
        assert len(got) == len(want)
        for k, (gv, wv) in enumerate(zip(got, want)):
            err = ((gv.double() - wv).abs().max() / wv.abs().max().clamp_min(1e-30)).item()
            assert gv.shape == wv.shape and err <= tol, (use_graph, k, err)
        if dt == torch.float32:
            # float32 vector and matrix states: the forward Scan AND the gradient Scan (mit-mot
            # [0, 1] -> [1], gate recomputation hoisted over the whole sequence) run as one
            # persistent kernel each
            assert list(ex.scan_modes.values()) == ["persistent", "persistent"], ex.scan_modes


def test_biased_rnn_bptt_against_torch_autograd():
    """A batched tanh RNN with a bias under aesara.grad at a real shape: the bias accumulator is
    rebuilt after the loop (fusion.push_out_accumulators), both Scans run persistent; loss and the
    gradients wrt W, U, b, h0 against torch.autograd of an fp64 restatement."""
    import torch
    from aesara_amd.executor import PlanExecutor
    T, B, D, H = 32, 48, 96, 128
    g = torch.Generator(device="cuda")
    g.manual_seed(23)
    x = torch.randn(T, B, D, device="cuda", generator=g) * 0.4
    h0 = torch.randn(B, H, device="cuda", generator=g) * 0.5
    W = torch.randn(D, H, device="cuda", generator=g) / np.sqrt(D)
    U = torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H)
    b = torch.randn(H, device="cuda", generator=g) * 0.1
    leaves = [t.double().requires_grad_(True) for t in (W, U, b, h0)]
    Wd, Ud, bd, h = leaves
    hs = []
    for t in range(T):
        h = torch.tanh(x[t].double() @ Wd + h @ Ud + bd)
        hs.append(h)
    loss = (hs[-1] ** 2).sum() + torch.stack(hs).mean()
    want = [loss.detach()] + [gr.detach() for gr in torch.autograd.grad(loss, leaves)]
    for use_graph in (False, True):
        ex = PlanExecutor(_plan("rnn_bias_bptt_b4_f32"), use_graph=use_graph)
        for _ in range(2):
            got = ex(x, h0, W, U, b)
        assert list(ex.scan_modes.values()) == ["persistent", "persistent"], ex.scan_modes
        for k, (gv, wv) in enumerate(zip(got, want)):
            err = ((gv.double() - wv).abs().max() / wv.abs().max().clamp_min(1e-30)).item()
            assert gv.shape == wv.shape and err <= 2e-4, (use_graph, k, err)


def test_fused_gate_lstm_bptt_against_torch_autograd():
    """The usual fused-gate LSTM under aesara.grad at a real T and B (H = 64 is baked into the
    golden's gate slices): forward AND gradient Scan on the persistent matrix kernel (column-slice
    splitting, accumulator push-out, block-wise gate gradients); loss and the gradients wrt W, U, b
    against torch.autograd of an fp64 restatement."""
    import torch
    from aesara_amd.executor import PlanExecutor
    T, B, D, H = 24, 40, 48, 64
    g = torch.Generator(device="cuda")
    g.manual_seed(29)
    x = torch.randn(T, B, D, device="cuda", generator=g) * 0.4
    h0 = torch.randn(B, H, device="cuda", generator=g) * 0.5
    c0 = torch.randn(B, H, device="cuda", generator=g) * 0.5
    W = torch.randn(D, 4 * H, device="cuda", generator=g) / np.sqrt(D)
    U = torch.randn(H, 4 * H, device="cuda", generator=g) / np.sqrt(H)
    b = torch.randn(4 * H, device="cuda", generator=g) * 0.1
    leaves = [t.double().requires_grad_(True) for t in (W, U, b)]
    Wd, Ud, bd = leaves
    h, c, hs, cs = h0.double(), c0.double(), [], []
    for t in range(T):
        gates = x[t].double() @ Wd + h @ Ud + bd
        i, f, o, gg = (gates[:, k * H:(k + 1) * H] for k in range(4))
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        hs.append(h)
        cs.append(c)
    loss = (hs[-1] ** 2).sum() + torch.stack(cs).mean()
    want = [loss.detach()] + [gr.detach() for gr in torch.autograd.grad(loss, leaves)]
    for use_graph in (False, True):
        ex = PlanExecutor(_plan("lstm_fused_bptt_h64_f32"), use_graph=use_graph)
        for _ in range(2):
            got = ex(x, h0, c0, W, U, b)
        assert list(ex.scan_modes.values()) == ["persistent", "persistent"], ex.scan_modes
        for k, (gv, wv) in enumerate(zip(got, want)):
            err = ((gv.double() - wv).abs().max() / wv.abs().max().clamp_min(1e-30)).item()
            assert gv.shape == wv.shape and err <= 2e-4, (use_graph, k, err)


def test_baseline_configs_against_the_reference_c_linker(tmp_path):
    """north_star: "results equal to the C linker within 1e-6 rel" — checked against the REFERENCE
    ITSELF, not a restatement: ``oracle/time_reference.py --dump-dir`` (child process, the
    reference's ``Mode("cvm","fast_run")`` from the packed overlay) evaluates cfg 2 / 1b / 3a / 3b at
    the full 4096 shapes, cfg 4 (B = 1 and B = 64) on a T = 64 sample and cfg 5 on N = 2^20 rows;
    ``tests/refcheck.py`` regenerates the same seeded inputs, runs the HIP path and compares
    every output (norm-wise relative error; |dx| / |ref| for scalars).  ``bench.py`` puts the
    same numbers on its line."""
    import os
    import subprocess
    import sys
    import ref_overlay
    if not ref_overlay.available():
        pytest.skip("no reference front end (oracle/_ref overlay not packed)")
    import refcheck
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("AESARA_FLAGS", None)
    cfgs = ["cfg2", "cfg1b", "cfg3a", "cfg3b", "cfg4_b1", "cfg4_b64", "cfg5"]
    p = subprocess.run([sys.executable, os.path.join(root, "oracle", "time_reference.py"),
                        "--budget", "0.01", "--configs", ",".join(cfgs), "--dump-dir", str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert "RESULT " in p.stdout, (p.stdout + p.stderr)[-2000:]
    got = refcheck.hip_vs_reference(str(tmp_path), cfgs)
    assert sorted(got) == sorted(cfgs), (sorted(got), p.stdout[-1500:])
    assert got["cfg1b"]["max"] == 0.0                          # an add is exact
    for k, v in got.items():
        assert v["ok"], (k, v)


@pytest.mark.parametrize("cfg", ["cfg4_b1", "cfg4_b64", "cfg5"])
def test_sampled_configs_at_full_shape_against_the_reference_c_linker(tmp_path, cfg):
    """The two configs whose TIMED rows are samples, at BASELINE's full shape against the
    reference itself: config 4 with all T = 512 steps (vector and B = 64 matrix state) and config
    5 with all N = 2^24 rows (16 GiB of X) — ``oracle/time_reference.py --full --budget 0``
    evaluates each exactly once with ``Mode("cvm","fast_run")``, the HIP path runs the same
    seeded inputs (SURVEY §8d "run it once").  ``bench.py`` puts the same numbers on its line
    (``vs_reference.full_shape``)."""
    import os
    import subprocess
    import sys
    import ref_overlay
    if not ref_overlay.available():
        pytest.skip("no reference front end (oracle/_ref overlay not packed)")
    if cfg == "cfg5":
        with open("/proc/meminfo") as f:
            avail = next(int(ln.split()[1]) for ln in f if ln.startswith("MemAvailable")) >> 20
        if avail < 40:
            pytest.skip("config 5 at N = 2^24 needs ~40 GiB of free host memory (%d GiB here)" % avail)
    import refcheck
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("AESARA_FLAGS", None)
    p = subprocess.run([sys.executable, os.path.join(root, "oracle", "time_reference.py"), "--full",
                        "--budget", "0", "--configs", cfg, "--dump-dir", str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert "RESULT " in p.stdout, (p.stdout + p.stderr)[-2000:]
    got = refcheck.hip_vs_reference(str(tmp_path), [cfg], full=True)
    assert list(got) == [cfg], (got, p.stdout[-1500:])
    v = got[cfg]
    assert v["full_shape"] and v["ok"], v
    shapes = v["input_shapes"]
    if cfg == "cfg5":
        assert shapes["X"] == [1 << 24, 256]
    else:
        assert shapes["x"][0] == 512 and shapes["x"][-1] == 1024


@pytest.mark.parametrize("shape,axis,dtype", [((4096, 4096), 0, "float64"), ((4096, 4096), 1, "float64"),
                                              ((16384, 4096), 0, "float32"), ((16384, 4096), 1, "float32"),
                                              ((256, 512, 256), 1, "float32")])
def test_big_axis_reductions_with_streaming_loads(shape, axis, dtype):
    """Axis reductions over operands of 96 MiB and more read them with non-temporal 16-byte loads
    (exec_elemwise.BIG_STREAM): sums against torch in float64 (CAReduce accumulates float32 sums in
    float64, tensor/elemwise.py:1495), twice (replay), and against the cached-load kernel."""
    import torch
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.plan import Node, Plan, Var
    nd = len(shape)
    pl = Plan("axisred", {0: Var(0, dtype, [None] * nd), 1: Var(1, dtype, [None] * (nd - 1))}, [0], [1],
              [Node("CAReduce", [0], [1], {"scalar_op": "add", "axis": [axis], "acc_dtype": "float64"})])
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    x = torch.randn(shape, dtype=getattr(torch, dtype), device="cuda", generator=g)
    want = x.double().sum(dim=axis)
    ex = PlanExecutor(pl, use_graph=True)
    tol = 1e-12 if dtype == "float64" else 2e-6
    for call in range(2):
        (got,) = ex(x)
        err = ((got.double() - want).norm() / want.norm()).item()
        assert err < tol, err
    ex.check()
