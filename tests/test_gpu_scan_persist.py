"""The Scan step loop as ONE persistent kernel (aesara_amd/scan_persist.py): parity with the
reference's outputs, with the launch-list path and with an fp64 restatement at the BASELINE
config-4 shape; replays (device-side tag epochs), workgroup geometries, fallbacks."""
import os

import numpy as np
import pytest

from golden_util import CASES, assert_matches, case_expected, case_inputs, case_plan

pytestmark = pytest.mark.gpu

PERSISTENT = ["cfg4_gru_b1_f32", "sp_gru_last_f32", "sp_lstm_vec_f32", "sp_rnn_proj_f32",
              "cfg4_gru_b8_f32", "gru_b1_f64",
              # taps older than -1 on a matrix state (round 4): registers of the tile element's owner
              "sm_taps_b16_f32", "sm_taps13_b32_f32"]


def _case(name):
    return next(c for c in CASES if c["name"] == name)


def _np(outs):
    return [o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o) for o in outs]


@pytest.mark.parametrize("name", PERSISTENT)
@pytest.mark.parametrize("use_graph", [False, True])
def test_persistent_path_is_taken_and_matches(name, use_graph):
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(4):                       # later calls: replay with advanced tag epochs
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"persistent call {it}")
    assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
    # identical to the launch-list path up to fp32 summation order
    E.TUNE["scan_persist"] = 0
    try:
        ex2 = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        ref = _np(ex2(*ins))
        assert all(v.startswith("launch-list") for v in ex2.scan_modes.values())
    finally:
        E.TUNE["scan_persist"] = 1
    tol = 1e-11 if name.endswith("f64") else 2e-5
    for g, r in zip(got, ref):
        np.testing.assert_allclose(g, r, rtol=tol, atol=tol)


@pytest.mark.parametrize("name", ["sp_taps_bptt_f64", "sm_taps_bptt_b16_f32", "scan_variant_1",
                                  "scan_variant_4", "scan_variant_7", "scan_variant_10"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_two_tap_bptt_runs_persistent(name, use_graph):
    """Round 4: the gradient Scan of a recurrence with taps [-1, -2] WITH products.  Two things
    kept it on the launch list: its mit-mot output has taps [0, 1, 2] -> [1, 2] (two states on one
    buffer: out-tap j of step i is tap j - 1 of step i + 1; scan/op.py:2379) and it accumulates
    the weight gradient inside the loop (a sit-sot Gemm / Ger per step that the reference's
    PushOutDot1 lifts for one tap only) — now ONE product over the stacked operands behind the
    loop (fusion.push_out_product_accumulators, decided per call: the buffer holds one row).
    Forward and gradient Scan both persistent; results = the reference's, = the launch-list path
    to summation order."""
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(3):
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"persistent call {it}")
    assert list(ex.scan_modes.values()) == ["persistent", "persistent"], ex.scan_modes
    E.TUNE["scan_persist"] = 0
    try:
        ex2 = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        ref = _np(ex2(*ins))
        assert all(v.startswith("launch-list") for v in ex2.scan_modes.values())
    finally:
        E.TUNE["scan_persist"] = 1
    tol = 2e-4 if name.endswith("f32") else 1e-10
    for g, r in zip(got, ref):
        np.testing.assert_allclose(g, r, rtol=tol, atol=tol)
    ex.check()


@pytest.mark.parametrize("name", ["scan_nitsot_value_and_its_view", "scan_nitsot_view_then_value"])
@pytest.mark.parametrize("persist", [1, 0])
@pytest.mark.parametrize("use_graph", [False, True])
def test_nitsot_value_and_its_view_both_handed_out(name, persist, use_graph):
    """A per-step vector handed out twice — as it is and as its [1, n] row view (DimShuffle 'x', 0) —
    next to the recurrent output: on the persistent kernel (the row view is the same rows under
    another shape) and on the launch list (each output row is a kernel's write target; one output
    being a view of another must not make them share one)."""
    from aesara_amd import executor as E
    c = _case(name)
    E.TUNE["scan_persist"] = persist
    try:
        ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        for it in range(3):
            got = _np(ex(*case_inputs(c)))
            assert_matches(c, got, case_expected(c), f"call {it}")
        assert all(v.startswith("persistent" if persist else "launch-list") for v in ex.scan_modes.values())
        ex.check()
    finally:
        E.TUNE["scan_persist"] = 1


def test_outside_the_class_falls_back():
    from aesara_amd.executor import PlanExecutor
    c = _case("sp_rnn_proj_narrow_f32")
    ex = PlanExecutor(case_plan(c))
    assert_matches(c, _np(ex(*case_inputs(c))), case_expected(c), "fallback")
    assert all(v.startswith("launch-list") for v in ex.scan_modes.values()), ex.scan_modes


def _gru_ref(x, h0, Ws):
    import torch
    Wz, Uz, Wr, Ur, Wh, Uh = [W.double() for W in Ws]
    h = h0.double()
    xd = x.double()
    xz, xr, xh = xd @ Wz, xd @ Wr, xd @ Wh
    hs = []
    for t in range(x.shape[0]):
        z = torch.sigmoid(xz[t] + h @ Uz)
        r = torch.sigmoid(xr[t] + h @ Ur)
        hh = torch.tanh(xh[t] + (r * h) @ Uh)
        h = (1 - z) * h + z * hh
        hs.append(h)
    return torch.stack(hs)


@pytest.mark.parametrize("T,H,rows", [(512, 1024, None), (512, 1024, 4), (64, 1000, None),
                                      (33, 1024, 16),     # all matrix rows in VGPRs
                                      (24, 2048, None),   # 48 MiB of weights: LDS + VGPRs
                                      (7, 260, None), (2, 64, None)])
def test_gru_full_size_all_steps_vs_fp64(T, H, rows, monkeypatch):
    """BASELINE config 4 (T=512, H=1024, fp32, B=1) and ragged relatives: EVERY step's state
    against an fp64 restatement (rel <= 1e-5 of the state's scale), eager and replayed."""
    import torch
    from aesara_amd.executor import PlanExecutor
    if rows:
        monkeypatch.setenv("AESARA_HIP_SCAN_ROWS", str(rows))
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    x = torch.randn(T, H, dtype=torch.float32, device="cuda", generator=g) * 0.1
    h0 = torch.randn(H, dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H)
          for _ in range(6)]
    ref = _gru_ref(x, h0, Ws)
    for use_graph in (False, True):
        ex = PlanExecutor(case_plan(_case("cfg4_gru_b1_f32")), use_graph=use_graph)
        for it in range(3):
            hs, hT = ex(x, h0, *Ws)
        assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
        assert hs.shape == (T, H)
        err = ((hs.double() - ref).abs().max() / ref.abs().max()).item()
        assert err <= 1e-5, err
        assert torch.equal(hT, hs[-1])


def test_persistent_replays_are_deterministic():
    import torch
    from aesara_amd.executor import PlanExecutor
    c = _case("sp_lstm_vec_f32")
    ex = PlanExecutor(case_plan(c), use_graph=True)
    ins = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in case_inputs(c)]
    first = [o.clone() for o in ex(*ins)]
    for _ in range(20):
        outs = ex(*ins)
        for a, b in zip(outs, first):
            assert torch.equal(a, b)


@pytest.mark.parametrize("batch", [0, 64])
def test_persistent_kernel_stress_many_replays_under_uneven_load(batch):
    """Hand-off protocol under load: 300 back-to-back replays of the config-4 loop (T = 64, 8192
    exchanges each) while a second stream keeps streaming a large buffer through the memory
    system; every evaluation must be bit-identical to the first (any stale / torn granule
    changes h_T)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    T, H = 64, 1024
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    shp = (T, H) if not batch else (T, batch, H)
    x = torch.randn(*shp, dtype=torch.float32, device="cuda", generator=g) * 0.1
    h0 = torch.randn(*shp[1:], dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H)
          for _ in range(6)]
    ex = PlanExecutor(case_plan(_case("cfg4_gru_b8_f32" if batch else "cfg4_gru_b1_f32")),
                      use_graph=True)
    hs0, _ = ex(x, h0, *Ws)
    assert list(ex.scan_modes.values()) == ["persistent"]
    if not batch:
        ref = _gru_ref(x, h0, Ws)
        assert ((hs0.double() - ref).abs().max() / ref.abs().max()).item() <= 1e-5
    first = hs0.clone()
    side = torch.cuda.Stream()
    big = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")     # 1 GiB
    bad = 0
    for it in range(300):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                big.add_(1.0)                  # HBM traffic + CUs taken by another stream
        hs, _ = ex(x, h0, *Ws)
        bad += int(not torch.equal(hs, first))
    torch.cuda.synchronize()
    ex.check()
    assert bad == 0, f"{bad} of 300 replays differ"


@pytest.mark.parametrize("T,H,B", [(512, 1024, 64), (40, 1024, 40), (9, 256, 5), (3, 64, 16)])
def test_gru_matrix_state_all_steps_vs_fp64(T, H, B):
    """The batch (matrix-state) class: BASELINE config 4 with B = 64 and ragged relatives (a last
    batch block with fewer than 16 rows) through the MFMA persistent kernel, EVERY step against
    an fp64 restatement, eager and replayed."""
    import torch
    from aesara_amd.executor import PlanExecutor
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.randn(T, B, H, dtype=torch.float32, device="cuda", generator=g) * 0.1
    h0 = torch.randn(B, H, dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H)
          for _ in range(6)]
    Wz, Uz, Wr, Ur, Wh, Uh = [W.double() for W in Ws]
    h = h0.double()
    ref = []
    for t in range(T):
        xt = x[t].double()
        z = torch.sigmoid(xt @ Wz + h @ Uz)
        r = torch.sigmoid(xt @ Wr + h @ Ur)
        hh = torch.tanh(xt @ Wh + (r * h) @ Uh)
        h = (1 - z) * h + z * hh
        ref.append(h)
    ref = torch.stack(ref)
    for use_graph in (False, True):
        ex = PlanExecutor(case_plan(_case("cfg4_gru_b8_f32")), use_graph=use_graph)
        for it in range(3):
            hs, hT = ex(x, h0, *Ws)
        assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
        assert hs.shape == (T, B, H)
        err = ((hs.double() - ref).abs().max() / ref.abs().max()).item()
        assert err <= 1e-5, err
        assert torch.equal(hT, hs[-1])


@pytest.mark.parametrize("T,H", [(64, 512), (17, 1024), (5, 26)])
def test_gru_float64_all_steps(T, H):
    """float64 state (Aesara's default floatX): values travel as two tagged granules (high / low
    word) and the loop runs in double — every step within 1e-12 of a torch fp64 restatement."""
    import torch
    from aesara_amd.executor import PlanExecutor
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    x = torch.randn(T, H, dtype=torch.float64, device="cuda", generator=g) * 0.1
    h0 = torch.randn(H, dtype=torch.float64, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float64, device="cuda", generator=g) / np.sqrt(H)
          for _ in range(6)]
    ref = _gru_ref(x, h0, Ws)
    for use_graph in (False, True):
        ex = PlanExecutor(case_plan(_case("gru_b1_f64")), use_graph=use_graph)
        for it in range(3):
            hs, hT = ex(x, h0, *Ws)
        assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
        err = ((hs - ref).abs().max() / ref.abs().max()).item()
        assert err <= 1e-12, err


@pytest.mark.parametrize("borrow", [False, True])
@pytest.mark.parametrize("batch", [0, 48])
def test_stateful_chunks_feed_the_last_state_back(borrow, batch):
    """A long sequence run as chunks, each call starting from the previous call's last state (a
    stateful RNN): the replayed persistent kernel sees a new ``x`` chunk and an ``h0`` that is the
    function's own previous result (borrowed: a range of its arena) on every call.  The chunked
    states must equal the one-shot run: exactly for the vector state (same kernel, same
    arithmetic), to fp32 round-off for the matrix state (the hoisted ``x @ W`` GEMMs pick their
    tiling by the row count, so the summation order differs between 768 and 4608 rows)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    T, H, CH = 96, 512, 16
    g = torch.Generator(device="cuda")
    g.manual_seed(21)
    shp = (T, batch, H) if batch else (T, H)
    x = torch.randn(*shp, dtype=torch.float32, device="cuda", generator=g) * 0.1
    h0 = torch.randn(*shp[1:], dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H)
          for _ in range(6)]
    name = "cfg4_gru_b8_f32" if batch else "cfg4_gru_b1_f32"
    whole, _ = PlanExecutor(case_plan(_case(name)))(x, h0, *Ws)
    ex = PlanExecutor(case_plan(_case(name)), use_graph=True, borrow=borrow)
    h, got = h0, []
    for c in range(T // CH):
        hs, h = ex(x[c * CH:(c + 1) * CH], h, *Ws)
        got.append(hs.clone())
    assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
    ex.check()
    if batch:
        assert torch.allclose(torch.cat(got), whole, rtol=1e-4, atol=2e-6)
    else:
        assert torch.equal(torch.cat(got), whole)


@pytest.mark.parametrize("name,dims", [("sp_lstm_vec_f32", {17: "T", 20: "D", 48: "H"}),
                                        ("sp_rnn_proj_f32", {19: "T", 12: "D", 32: "H"}),
                                        ("sp_gru_last_f32", {21: "T", 52: "H"}),
                                        ("gru_b1_f64", None),
                                        ("gru_bptt_b1_f32", {12: "T", 20: "H"}),
                                        ("lstm_bptt_vec_f32", {9: "T", 12: "D", 16: "H"}),
                                        ("gru_bptt_b4_f32", {10: "T", 4: "B", 16: "H"}),
                                        ("cfg4_gru_b8_f32", {16: "T", 8: "B", 64: "H"})])
def test_persistent_scans_on_random_extents(name, dims):
    """The golden recurrences (LSTM with two states, RNN with a per-step projection output, GRU
    that returns only the last state, float64 GRU) at random extents — ragged state sizes (12,
    68, 100, 260, 516 ...), odd ones whose weight rows are zero-padded (31, 7), single steps, inputs
    wider than the state: the one-kernel loop against the launch-list path (same arithmetic up to
    summation order)."""
    import torch
    from aesara_amd import executor as E
    c = _case(name)
    base = case_inputs(c)
    if dims is None:
        dims = {base[0].shape[0]: "T", base[1].shape[0]: "H"}
    rng = np.random.default_rng(31)
    f64 = name.endswith("f64")
    for trial in range(10):
        batched = "B" in dims.values()
        val = {"T": int(rng.choice([1, 2, 3, 9, 30])), "D": int(rng.choice([4, 7, 36, 72])),
               "H": int(rng.choice([64, 100, 129, 320, 37] if batched else [4, 12, 31, 68, 100, 260, 384, 516])),
               "B": int(rng.choice([1, 5, 16, 40]))}
        ins = []
        for a in base:
            shp = tuple(val[dims[n]] for n in a.shape)
            scale = 0.5 if len(shp) == 1 else 1.0 / np.sqrt(shp[0])
            ins.append((rng.standard_normal(shp) * scale).astype(a.dtype))
        dev = [torch.from_numpy(a).cuda() for a in ins]
        ex = E.PlanExecutor(case_plan(c), use_graph=True)
        for _ in range(2):
            got = _np(ex(*dev))
        # weight rows are read as 16-byte vectors: contraction lengths that are not a multiple of
        # 4 (float32) / 2 (float64) elements get zero-padded copies (still the one-kernel loop)
        vec_ok = True
        if batched:       # matrix state: one 16 x 16 tile per CU over the width padded to 64
            vec_ok = -(-val["B"] // 16) * (-(-val["H"] // 64) * 4) <= 256
        if vec_ok and val["T"] >= 2:
            assert set(ex.scan_modes.values()) == {"persistent"}, (val, ex.scan_modes)
        ex.check()
        E.TUNE["scan_persist"] = 0
        try:
            ref = _np(E.PlanExecutor(case_plan(c))(*dev))
        finally:
            E.TUNE["scan_persist"] = 1
        tol = 1e-11 if f64 else (3e-4 if "bptt" in name else 3e-5)
        for g, r in zip(got, ref):
            assert g.shape == r.shape, (val, g.shape, r.shape)
            np.testing.assert_allclose(g, r, rtol=tol, atol=tol * max(1.0, float(np.abs(r).max(initial=0))),
                                       err_msg=str(val))


@pytest.mark.parametrize("batch", [0, 32])
def test_varying_chunk_lengths_share_one_exchange_workspace(batch):
    """Calls with different step counts (16, 20, 16, 7, 2, 33 ...) are different replay signatures
    of the SAME persistent kernel and exchange workspace (the tag epoch advances by each call's
    step count): the chained states must equal the one-shot run."""
    import torch
    from aesara_amd.executor import PlanExecutor
    H = 256
    lens = [16, 20, 16, 7, 2, 33, 20, 2]
    T = sum(lens)
    g = torch.Generator(device="cuda")
    g.manual_seed(22)
    shp = (T, batch, H) if batch else (T, H)
    x = torch.randn(*shp, dtype=torch.float32, device="cuda", generator=g) * 0.1
    h0 = torch.randn(*shp[1:], dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H)
          for _ in range(6)]
    name = "cfg4_gru_b8_f32" if batch else "cfg4_gru_b1_f32"
    whole, _ = PlanExecutor(case_plan(_case(name)))(x, h0, *Ws)
    ex = PlanExecutor(case_plan(_case(name)), use_graph=True)
    for rep in range(2):
        h, got, t0 = h0, [], 0
        for n in lens:
            hs, h = ex(x[t0:t0 + n], h, *Ws)
            got.append(hs)
            t0 += n
        ex.check()
        cat = torch.cat(got)
        if batch:
            assert torch.allclose(cat, whole, rtol=1e-4, atol=2e-6), rep
        else:
            assert torch.equal(cat, whole), rep
    assert set(ex.scan_modes.values()) == {"persistent"}, ex.scan_modes


@pytest.mark.parametrize("name", ["gru_bptt_b1_f32", "lstm_bptt_vec_f32", "scan_grad_last_state_f32",
                                  "lstm_fused_vec_bptt_f32", "lstm_fused_bptt_h64_f32", "rnn_bias_bptt_b4_f32"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_gradient_scans_run_persistent_and_match(name, use_graph):
    """aesara.grad through a recurrence: the forward Scan and the gradient Scan (mit-mot groups
    [0, 1] -> [1] for every propagated state; the gate recomputation from stored states lifted
    out of the step as whole-sequence Elemwise kernels and GEMMs) both take the one-kernel loop;
    loss and gradients against the reference's outputs, and against the launch-list path."""
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(3):
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"persistent gradient call {it}")
    assert set(ex.scan_modes.values()) == {"persistent"} and len(ex.scan_modes) == 2, ex.scan_modes
    E.TUNE["scan_persist"] = 0
    try:
        ref = _np(E.PlanExecutor(case_plan(c))(*ins))
    finally:
        E.TUNE["scan_persist"] = 1
    for g, r in zip(got, ref):
        np.testing.assert_allclose(g, r, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("batch", [0, 16])
def test_sgd_loop_over_bptt_plan_replay_equals_eager(batch):
    """Five SGD steps on the GRU weights through the loss-and-gradients plan: every call gets NEW
    weight tensors (the replayed launch lists — two persistent kernels, whole-sequence GEMMs —
    are rebound to them); the loss trajectory and the final weights of the replayed executor
    equal the eager executor's to fp32 round-off."""
    import torch
    from aesara_amd.executor import PlanExecutor
    T, H = 24, 128
    name = "gru_bptt_b4_f32" if batch else "gru_bptt_b1_f32"
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    shp = (T, batch, H) if batch else (T, H)
    x = torch.randn(*shp, device="cuda", generator=g) * 0.3
    h0 = torch.randn(*shp[1:], device="cuda", generator=g) * 0.5
    W0 = [torch.randn(H, H, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    traj = {}
    for mode in (False, True):
        ex = PlanExecutor(case_plan(_case(name)), use_graph=mode)
        Ws, losses = [w.clone() for w in W0], []
        for it in range(5):
            outs = ex(x, h0, *Ws)
            losses.append(outs[0].item())
            Ws = [w - 0.05 * gw for w, gw in zip(Ws, outs[1:7])]     # new tensors every step
        ex.check()
        assert set(ex.scan_modes.values()) == {"persistent"}, ex.scan_modes
        traj[mode] = (losses, Ws)
    assert traj[False][0][-1] < traj[False][0][0]          # the loss goes down
    np.testing.assert_allclose(traj[True][0], traj[False][0], rtol=2e-5)
    for a, b in zip(traj[True][1], traj[False][1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("T,B,D", [(48, 64, 96), (7, 5, 20)])
def test_fused_gate_lstm_runs_persistent(T, B, D):
    """The usual LSTM step — one product for the four gates, sliced by columns — is split per gate
    (fusion.split_column_slices: column views of W / U / b) and runs as the matrix-state persistent
    kernel with ONE hand-off per step (h; the cell state stays in the element owners' registers).
    Every hidden state and the last cell state against an fp64 restatement."""
    import torch
    from aesara_amd.executor import PlanExecutor
    H = 64
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    x = torch.randn(T, B, D, device="cuda", generator=g) * 0.5
    h0 = torch.randn(B, H, device="cuda", generator=g) * 0.5
    c0 = torch.randn(B, H, device="cuda", generator=g) * 0.5
    W = torch.randn(D, 4 * H, device="cuda", generator=g) / np.sqrt(D)
    U = torch.randn(H, 4 * H, device="cuda", generator=g) / np.sqrt(H)
    b = torch.randn(4 * H, device="cuda", generator=g) * 0.1
    h, c, hs = h0.double(), c0.double(), []
    for t in range(T):
        gates = x[t].double() @ W.double() + h @ U.double() + b.double()
        i, f, o, gg = (gates[:, k * H:(k + 1) * H] for k in range(4))
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        hs.append(h)
    want_hs, want_c = torch.stack(hs), c
    for use_graph in (False, True):
        ex = PlanExecutor(case_plan(_case("lstm_fused_fwd_f32")), use_graph=use_graph)
        for _ in range(2):
            got_hs, got_c = ex(x, h0, c0, W, U, b)
        assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
        ex.check()
        for gv, wv in ((got_hs, want_hs), (got_c, want_c)):
            err = ((gv.double() - wv).abs().max() / wv.abs().max()).item()
            assert gv.shape == wv.shape and err <= 2e-5, (use_graph, err)


@pytest.mark.parametrize("T,D", [(64, 40), (3, 7)])
def test_fused_gate_lstm_vector_state_runs_persistent(T, D):
    """The fused-gate LSTM step on a vector state (one sequence): the 4H-long ``Gemv`` chain is
    split per gate (row blocks of W.T / U.T), the ``x_t`` products are lifted over the sequence,
    the loop exchanges only h — all states against an fp64 restatement."""
    import torch
    from aesara_amd.executor import PlanExecutor
    H = 48
    g = torch.Generator(device="cuda")
    g.manual_seed(19)
    x = torch.randn(T, D, device="cuda", generator=g) * 0.5
    h0 = torch.randn(H, device="cuda", generator=g) * 0.5
    c0 = torch.randn(H, device="cuda", generator=g) * 0.5
    W = torch.randn(D, 4 * H, device="cuda", generator=g) / np.sqrt(D)
    U = torch.randn(H, 4 * H, device="cuda", generator=g) / np.sqrt(H)
    b = torch.randn(4 * H, device="cuda", generator=g) * 0.1
    h, c, hs = h0.double(), c0.double(), []
    for t in range(T):
        gates = x[t].double() @ W.double() + h @ U.double() + b.double()
        i, f, o, gg = (gates[k * H:(k + 1) * H] for k in range(4))
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        hs.append(h)
    want_hs, want_c = torch.stack(hs), c
    for use_graph in (False, True):
        ex = PlanExecutor(case_plan(_case("lstm_fused_vec_f32")), use_graph=use_graph)
        for _ in range(2):
            got_hs, got_c = ex(x, h0, c0, W, U, b)
        if T >= 2:
            assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
        ex.check()
        for gv, wv in ((got_hs, want_hs), (got_c, want_c)):
            err = ((gv.double() - wv).abs().max() / wv.abs().max()).item()
            assert gv.shape == wv.shape and err <= 2e-5, (use_graph, err)


def test_persistent_scan_next_to_a_saturating_stream_and_occupancy_refusal():
    """VERDICT r2 #6c.  (1) A second stream keeps every CU busy with long kernels while the
    persistent Scan kernel (whose workgroups wait for each other) runs: the answer is right — or
    the call raises promptly (bounded spins -> error word -> RuntimeError in THIS call).
    (2) A grid that cannot be co-resident is refused BEFORE launch (occupancy query) and the
    launch-list path runs instead."""
    import torch
    from golden_util import CASES, case_expected, case_inputs, case_plan
    from aesara_amd.executor import PlanExecutor, _Kernels
    c = next(c for c in CASES if c["name"] == "sp_gru_last_f32")
    plan = case_plan(c)
    ins = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in case_inputs(c)]
    want = case_expected(c)
    ex = PlanExecutor(plan, use_graph=False)
    ex(*ins)
    assert list(ex.scan_modes.values()) == ["persistent"]
    side = torch.cuda.Stream()
    big = torch.randn(8192, 8192, device="cuda")
    ok = err = 0
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(4):
                big = torch.tanh(big @ big * 1e-4)            # ~1.1 TFLOP each: the device is full
        try:
            outs = ex(*ins)
            for o, w in zip(outs, want):
                np.testing.assert_allclose(o.cpu().numpy(), w, rtol=c["rtol"], atol=c["atol"])
            ok += 1
        except RuntimeError as e:
            assert "persistent Scan kernel" in str(e)
            err += 1
        side.synchronize()
    assert ok + err == 6 and ok >= 1
    outs = ex(*ins)                                            # and afterwards all is well
    for o, w in zip(outs, want):
        np.testing.assert_allclose(o.cpu().numpy(), w, rtol=c["rtol"], atol=c["atol"])
    # (2) pretend the device holds fewer workgroups than the grid needs
    ex2 = PlanExecutor(plan, use_graph=False)
    saved = dict(_Kernels.cache)
    try:
        for k in list(_Kernels.cache):
            if isinstance(k, tuple) and k and k[0] == "occ":
                _Kernels.cache[k] = 1
        outs = ex2(*ins)
        (mode,) = ex2.scan_modes.values()
        assert mode.startswith("launch-list: grid of") and "co-resident" in mode
        for o, w in zip(outs, want):
            np.testing.assert_allclose(o.cpu().numpy(), w, rtol=c["rtol"], atol=c["atol"])
    finally:
        _Kernels.cache.clear()
        _Kernels.cache.update(saved)


@pytest.mark.parametrize("T,H,B", [(7, 64, 16), (33, 128, 32), (12, 256, 48), (64, 1024, 64),
                                   # round 4 (hand-offs re-scheduled): loops shorter than the schedule's
                                   # look-ahead, 8 fragments per product (the staged-operand product
                                   # split without a look inside it)
                                   (2, 512, 32), (2, 1024, 16), (3, 1024, 64), (9, 512, 64),
                                   # more 16 x 16 tiles than CUs: consecutive launches over batch slices
                                   (5, 1024, 128), (4, 1024, 256), (6, 512, 160)])
def test_sequence_products_inside_the_persistent_loop(T, H, B, monkeypatch):
    """Round 3: the ``x_t @ W`` products of a batched recurrence are computed INSIDE the persistent
    matrix kernel (fragment-ordered x, weight columns in LDS — as many products as fit, the others
    stay up front: two of three at H = 1024 —, accumulated into the recurrent products'
    accumulators) instead of as a GEMM over the whole sequence up front.
    Same results as with the products up front (AESARA_HIP_SM_XFOLD=0) up to fp32 summation
    order, every step within 1e-5 of fp64, eager and replayed; an ineligible batch (B % 16 != 0)
    keeps the up-front form."""
    import torch
    from aesara_amd.executor import PlanExecutor
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    x = torch.randn(T, B, H, dtype=torch.float32, device="cuda", generator=g) * 0.3
    h0 = torch.randn(B, H, dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    Wz, Uz, Wr, Ur, Wh, Uh = [W.double() for W in Ws]
    h, ref = h0.double(), []
    for t in range(T):
        xt = x[t].double()
        z = torch.sigmoid(xt @ Wz + h @ Uz)
        r = torch.sigmoid(xt @ Wr + h @ Ur)
        h = (1 - z) * h + z * torch.tanh(xt @ Wh + (r * h) @ Uh)
        ref.append(h)
    ref = torch.stack(ref)
    plan = case_plan(_case("cfg4_gru_b8_f32"))
    res = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("AESARA_HIP_SM_XFOLD", fold)
        for use_graph in (False, True):
            ex = PlanExecutor(plan, use_graph=use_graph)
            for _ in range(3):
                hs, hT = ex(x, h0, *Ws)
            assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
            notes = list(ex.scan_notes.values())
            assert notes == (["sequence products in the loop"] if fold == "1" else []), notes
            err = ((hs.double() - ref).abs().max() / ref.abs().max()).item()
            assert err <= 1e-5, (fold, use_graph, err)
            res[(fold, use_graph)] = hs.clone()
            ex.check()
    assert torch.allclose(res[("1", False)], res[("0", False)], rtol=2e-5, atol=2e-6)
    assert torch.equal(res[("1", False)], res[("1", True)])
    monkeypatch.setenv("AESARA_HIP_SM_XFOLD", "1")
    ex = PlanExecutor(plan)
    ex(x[:, :B - 3].contiguous(), h0[:B - 3].contiguous(), *Ws)             # ragged batch: up front
    assert list(ex.scan_modes.values()) == ["persistent"] and not ex.scan_notes


@pytest.mark.parametrize("T,H,B,chunks", [(6, 1024, 200, "1"), (6, 1024, 128, "0"), (5, 512, 300, "1"),
                                          (5, 1024, 136, "2")])
def test_batches_beyond_one_tile_per_workgroup(T, H, B, chunks, monkeypatch):
    """B * H / 256 tiles > CUs: consecutive launches of the one-block-per-workgroup kernel over slices
    of the batch (SM_BATCH_CHUNKS, the default for forward Scans; a ragged last slice included) or
    one launch with 2 / 4 / 8 blocks per workgroup (=0) — every step within 1e-5 of fp64 either way."""
    import torch
    from aesara_amd.executor import PlanExecutor
    monkeypatch.setenv("AESARA_HIP_SM_BATCH_CHUNKS", chunks)
    g = torch.Generator(device="cuda")
    g.manual_seed(13)
    x = torch.randn(T, B, H, dtype=torch.float32, device="cuda", generator=g) * 0.3
    h0 = torch.randn(B, H, dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    Wz, Uz, Wr, Ur, Wh, Uh = [W.double() for W in Ws]
    h, ref = h0.double(), []
    for t in range(T):
        xt = x[t].double()
        z = torch.sigmoid(xt @ Wz + h @ Uz)
        r = torch.sigmoid(xt @ Wr + h @ Ur)
        h = (1 - z) * h + z * torch.tanh(xt @ Wh + (r * h) @ Uh)
        ref.append(h)
    ref = torch.stack(ref)
    for use_graph in (False, True):
        ex = PlanExecutor(case_plan(_case("cfg4_gru_b8_f32")), use_graph=use_graph)
        for _ in range(3):
            hs, hT = ex(x, h0, *Ws)
        if chunks != "0" or B % 32 == 0:
            assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
        err = ((hs.double() - ref).abs().max() / ref.abs().max()).item()
        assert err <= 1e-5, (use_graph, err)
        assert torch.equal(hT, hs[-1])
        ex.check()


@pytest.mark.parametrize("knobs_", [{"AESARA_HIP_SM_XREG": "1"}, {"AESARA_HIP_SM_POLLS": "4"},
                                    {"AESARA_HIP_SM_INIT": "branch"}, {"AESARA_HIP_SM_XRELOAD": "early"},
                                    {"AESARA_HIP_SM_EPRE": "0", "AESARA_HIP_SM_XSPLIT": "0", "AESARA_HIP_SM_XPRE": "0"},
                                    {"AESARA_HIP_SM_ACKFILL": "0", "AESARA_HIP_SM_LOOK": "0", "AESARA_HIP_SM_PIN": "1"}])
def test_opt_in_schedules_of_the_matrix_kernel_compute_the_same(knobs_, monkeypatch):
    """The switches DESIGN §3.3b (round 4) measured and left off (third sequence product with its
    weight columns in registers, several polls in flight, the step-0 path in the loop, the next x
    requested early, no split products, the round-3 schedule) still compute the recurrence: every
    step within 1e-5 of fp64 at the benchmark's width."""
    import torch
    from aesara_amd.executor import PlanExecutor
    for k, v in knobs_.items():
        monkeypatch.setenv(k, v)
    T, H, B = 11, 1024, 64
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    x = torch.randn(T, B, H, dtype=torch.float32, device="cuda", generator=g) * 0.3
    h0 = torch.randn(B, H, dtype=torch.float32, device="cuda", generator=g) * 0.5
    Ws = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H) for _ in range(6)]
    Wz, Uz, Wr, Ur, Wh, Uh = [W.double() for W in Ws]
    h, ref = h0.double(), []
    for t in range(T):
        xt = x[t].double()
        z = torch.sigmoid(xt @ Wz + h @ Uz)
        r = torch.sigmoid(xt @ Wr + h @ Ur)
        h = (1 - z) * h + z * torch.tanh(xt @ Wh + (r * h) @ Uh)
        ref.append(h)
    ref = torch.stack(ref)
    ex = PlanExecutor(case_plan(_case("cfg4_gru_b8_f32")))
    for _ in range(2):
        hs, hT = ex(x, h0, *Ws)
    assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
    err = ((hs.double() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 1e-5, (knobs_, err)
    ex.check()


@pytest.mark.parametrize("name", ["xfold_gru_b16_f32", "xfold_rnn_b16_f32"])
def test_golden_recurrences_with_the_products_in_the_loop(name):
    """The two golden recurrences whose shapes let the kernel take the sequence products in
    (batch % 16 == 0, state % 64 == 0) against the REFERENCE's outputs: the GRU spreads three
    products over its two windows (one of them computed a step ahead, from x_{t+1}); the Elman
    step has a single fetching phase, so its product runs a whole step ahead."""
    from aesara_amd.executor import PlanExecutor
    c = _case(name)
    plan, ins, want = case_plan(c), case_inputs(c), case_expected(c)
    for use_graph in (False, True):
        ex = PlanExecutor(plan, use_graph=use_graph)
        for _ in range(2):
            got = ex(*ins)
        assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
        assert list(ex.scan_notes.values()) == ["sequence products in the loop"], ex.scan_notes
        assert_matches(c, [g_.cpu().numpy() for g_ in got], want)
        ex.check()


@pytest.mark.parametrize("T,H,B", [(5, 64, 16), (40, 256, 32), (24, 1024, 64)])
def test_elman_sequence_product_a_step_ahead(T, H, B):
    """h_t = tanh(x_t W + h_{t-1} U) at sizes up to the benchmark's: the product for step t + 1 is
    accumulated in the window of step t (x_{t+2} requested behind it); every step within 1e-5 of
    fp64."""
    import torch
    from aesara_amd.executor import PlanExecutor
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.randn(T, B, H, dtype=torch.float32, device="cuda", generator=g) * 0.5
    h0 = torch.randn(B, H, dtype=torch.float32, device="cuda", generator=g) * 0.5
    W, U = [torch.randn(H, H, dtype=torch.float32, device="cuda", generator=g) / np.sqrt(H) for _ in range(2)]
    h, ref = h0.double(), []
    for t in range(T):
        h = torch.tanh(x[t].double() @ W.double() + h @ U.double())
        ref.append(h)
    ref = torch.stack(ref)
    ex = PlanExecutor(case_plan(_case("xfold_rnn_b16_f32")))
    for _ in range(2):
        hs, hT = ex(x, h0, W, U)
    assert list(ex.scan_notes.values()) == ["sequence products in the loop"], (ex.scan_modes, ex.scan_notes)
    err = ((hs.double() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 1e-5, err
    assert torch.equal(hT, hs[-1])


# ---- full reductions inside the step of the vector-state kernel (round 5) ------------------------
RED_BOTH = ["scan_variant_0", "scan_variant_3", "scan_variant_6", "scan_variant_9",        # forward + gradient Scan
            # (gradient Scans with mit-mot taps [0, 1, 3] -> [1, 3]: the out-tap read again two steps later
            # stays in the row owner's registers, scan_persist.analyze)
            "scan_variant_2", "scan_variant_5", "scan_variant_8", "scan_variant_11"]
RED_FWD = []
RED_ONE = ["scan_while_nitsot_matrix", "scan_red_rnn_normalised", "scan_red_rnn_until_f32"]


@pytest.mark.parametrize("name", RED_BOTH + RED_FWD + RED_ONE)
@pytest.mark.parametrize("use_graph", [False, True])
def test_steps_with_reductions_run_in_the_persistent_kernel(name, use_graph):
    """Steps with full reductions next to the recurrent dot (``(h ** 2).sum()``, ``x_t.max()``, a state
    divided by its own norm, a do-while on ``h.max()``; scan_perform.pyx:309-541, :424-426): the
    reduced vector is exchanged like the state, every workgroup folds it in one fixed order.
    Results = the reference's outputs = the launch-list path (at the case's tolerance: other
    summation order)."""
    from aesara_amd import executor as E
    c = _case(name)
    ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
    ins = case_inputs(c)
    for it in range(4):                       # later calls: advanced tag epochs (and replays)
        got = _np(ex(*ins))
        assert_matches(c, got, case_expected(c), f"persistent call {it}")
    modes = list(ex.scan_modes.values())
    if name in RED_FWD:
        assert modes.count("persistent") >= 1, ex.scan_modes
    else:
        assert all(m == "persistent" for m in modes), ex.scan_modes
    E.TUNE["scan_persist"] = 0
    try:
        ex2 = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        ref = _np(ex2(*ins))
        assert all(v.startswith("launch-list") for v in ex2.scan_modes.values())
    finally:
        E.TUNE["scan_persist"] = 1
    assert_matches(c, got, ref, "against the launch-list path")
    ex.check()


@pytest.mark.parametrize("scale", [0.01, 0.3, 0.5, 1.0, 8.0])
def test_vector_state_do_while_trip_counts(scale):
    """h <- 0.9 h + 0.5 tanh(W h + x_t) until max(h) > 1.25 (golden scan_red_rnn_until_f32's step):
    inputs scaled so that the loop never stops, stops late, early, after one step — trip count and
    truncated outputs against a NumPy loop in float64 (float32 tolerance)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(3)
    T, H = 40, 96
    x = (rng.standard_normal((T, H)) * scale).astype("float32")
    W = (rng.standard_normal((H, H)) * 0.1).astype("float32")
    h0 = np.zeros(H, dtype="float32")
    h, rows, mx = h0.astype("float64"), [], []
    for t in range(T):
        h = 0.9 * h + 0.5 * np.tanh(W.astype("float64") @ h + x[t])
        rows.append(h.copy())
        mx.append(h.max())
        if h.max() > 1.25:
            break
    if any(abs(m - 1.25) < 1e-3 for m in mx):
        pytest.skip("a maximum within float32 noise of the threshold")
    ex = PlanExecutor(case_plan(_case("scan_red_rnn_until_f32")), use_graph=True)
    for call in range(2):
        hs, ms, count = _np(ex(torch.from_numpy(x).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(h0).cuda()))
        assert int(count) == len(rows) and hs.shape == (len(rows), H) and ms.shape == (len(rows),)
        np.testing.assert_allclose(hs, np.array(rows), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(ms, np.array(mx), rtol=2e-4, atol=2e-5)
    assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
    ex.check()


@pytest.mark.parametrize("use_graph", [False, True])
def test_embedding_lookup_inside_the_step_leaves_the_loop(use_graph):
    """``E[idx_t]`` inside a recurrent step (scalar index, vector state; index vector, float32 matrix
    state) is restated over whole sequences in front of the Scan (fusion.push_out_sequence_glue): the
    forward loops run on the persistent kernels, results = the reference's, and at T = 300 with a
    10^4-row table = NumPy."""
    import torch
    from aesara_amd import executor as E
    for name in ("scan_embedding_lookup_in_step", "scan_embedding_lookup_batch_f32"):
        c = _case(name)
        ex = E.PlanExecutor(case_plan(c), use_graph=use_graph)
        for it in range(2):
            assert_matches(c, _np(ex(*case_inputs(c))), case_expected(c), f"{name} call {it}")
        # (forward AND gradient Scan: dE[idx_t] += delta_t leaves its loop as one scatter-add)
        assert set(ex.scan_modes.values()) == {"persistent"} and len(ex.scan_modes) == 2, ex.scan_modes
    rng = np.random.default_rng(5)
    T, B, V, H = 300, 32, 10000, 128
    idx = rng.integers(0, V, (T, B))
    Em = (rng.standard_normal((V, H)) * 0.5).astype("float32")
    U = (rng.standard_normal((H, H)) / np.sqrt(H)).astype("float32")
    h0 = (rng.standard_normal((B, H)) * 0.1).astype("float32")
    ex = E.PlanExecutor(case_plan(_case("scan_embedding_lookup_batch_f32")), use_graph=use_graph)
    hs, dU, dE = _np(ex(*(torch.from_numpy(a).cuda() for a in (idx, Em, U, h0))))
    h, want = h0.astype(np.float64), []
    U64 = U.astype(np.float64)
    for t in range(T):
        h = np.tanh(Em[idx[t]].astype(np.float64) + h @ U64)
        want.append(h)
    np.testing.assert_allclose(hs, np.stack(want), rtol=2e-4, atol=2e-5)
    # cost = sum(h_T^2): back through time in float64; the table's gradient is a scatter of the
    # per-step pre-activation gradients (every index repeats ~ T * B / V times)
    g = 2.0 * want[-1]
    dU_w, dE_w = np.zeros((H, H)), np.zeros((V, H))
    for t in range(T - 1, -1, -1):
        da = g * (1.0 - want[t] ** 2)
        hp = want[t - 1] if t else h0.astype(np.float64)
        dU_w += hp.T @ da
        np.add.at(dE_w, idx[t], da)
        g = da @ U64.T
    np.testing.assert_allclose(dU, dU_w, rtol=2e-3, atol=2e-5 * np.abs(dU_w).max())
    np.testing.assert_allclose(dE, dE_w, rtol=2e-3, atol=2e-5 * np.abs(dE_w).max())
    assert set(ex.scan_modes.values()) == {"persistent"}, ex.scan_modes
    bad = np.array(idx, copy=True)
    bad[7, 3] = V                      # out of range at step 7: the reference's IndexError
    with pytest.raises(IndexError):
        ex(*(torch.from_numpy(a).cuda() for a in (bad, Em, U, h0)))
        ex.check()
