"""Host-side logic of the HIP linker, exercised without a GPU: plan (de)serialisation, fusion,
stride arithmetic, code generation and — through the executor's dry-run mode — the complete
shape/stride/kernel-selection path for every golden case."""
import json

import numpy as np
import pytest

from golden_util import CASE_IDS, CASES, case_expected, case_inputs, case_plan

from aesara_amd import codegen as cg
from aesara_amd import hostops
from aesara_amd.device import DevArray
from aesara_amd.executor import PlanExecutor, collapse_dims
from aesara_amd.fusion import build_steps
from aesara_amd.plan import Plan


def _case(name):
    return next(c for c in CASES if c["name"] == name)


def test_plan_json_roundtrip():
    for c in CASES[::7]:
        p = case_plan(c)
        p2 = Plan.loads(p.dumps())
        assert json.dumps(p.to_json(), sort_keys=True) == json.dumps(p2.to_json(), sort_keys=True)


def test_cfg2_fuses_to_one_kernel():
    """BASELINE config 2: DimShuffle x2, square(sigma), Composite, Sum -> 2 views + ONE fused
    Elemwise+CAReduce step whose only array operand is x (the reference keeps a
    128 MiB intermediate between Composite and Sum)."""
    steps = build_steps(case_plan(_case("cfg2_gauss_sum")))
    kinds = [s.kind for s in steps]
    assert kinds.count("reduce") == 1 and kinds.count("elemwise") == 0
    red = next(s for s in steps if s.kind == "reduce")
    assert red.outputs == [] and len(red.inputs) == 3      # x, mu(1,1), sigma(1,1); -0.5 inlined
    ops = [n["op"] for n in red.scalar["nodes"]]
    assert ops == ["sqr", "sub", "sqr", "mul", "true_div", "exp"]


def test_cfg5_fuses_to_single_pass_rowpass():
    """BASELINE config 5 (14 reference nodes, X read twice): ONE row-program kernel that reads X
    once + folds of its per-workgroup partials; the unfused steps are kept as the fallback."""
    steps = build_steps(case_plan(_case("cfg5_logistic")))
    rp = [s for s in steps if s.kind == "rowpass"]
    assert len(rp) == 1 and not any(s.kind in ("gemv_epi", "reduce") for s in steps)
    rp = rp[0]
    assert rp.outputs == []                       # z and r never leave the kernel
    assert len(rp.extra["reds"]) == 2             # logp and d/db partial sums (fp64)
    assert [q.kind for q in rp.post] == ["reduce", "elemwise", "reduce", "reduce"]
    reds = [s for s in rp.fallback if s.kind == "reduce"]
    # in the unfused form r feeds both Gemv and Sum: it must stay materialised there
    assert sorted(len(s.outputs) for s in reds) == [0, 1]


def test_softmax_and_layernorm_fuse_to_one_rowchain():
    """A last-axis reduction chain (Softmax: max -> exp-sum -> scale; a hand-written layer norm:
    mean -> centred square mean -> normalise) becomes ONE row-chain step; nothing between the
    reductions is materialised unless the graph asks for it."""
    steps = build_steps(case_plan(_case("softmax_rows_f32")))
    rc = [s for s in steps if s.kind == "rowchain"]
    assert len(rc) == 1 and not any(s.kind in ("reduce", "elemwise") for s in steps)
    assert [m["reduce"]["op"] if m["reduce"] else None for m in rc[0].extra["members"]] == \
        ["maximum", "add", None]
    assert len(rc[0].outputs) == 1 and len(rc[0].fallback) == 5
    steps = build_steps(case_plan(_case("layernorm_float32")))
    rc = [s for s in steps if s.kind == "rowchain"]
    assert len(rc) == 1 and not any(s.kind == "reduce" for s in steps)
    assert len(rc[0].outputs) == 3          # y, mu and var are graph outputs


def test_unfused_steps_mirror_plan():
    p = case_plan(_case("cfg2_gauss_sum"))
    assert len(build_steps(p, fuse=False)) == len(p.nodes)


def test_collapse_dims():
    assert collapse_dims([4096, 4096], [[4096, 1], [0, 0]]) == ([16777216], [[1], [0]])
    assert collapse_dims([3, 5], [[5, 1], [0, 1]]) == ([3, 5], [[5, 1], [0, 1]])
    assert collapse_dims([2, 3, 4], [[12, 4, 1], [1, 2, 6]]) == ([2, 3, 4], [[12, 4, 1], [1, 2, 6]])
    assert collapse_dims([1, 1], [[7, 9]]) == ([1], [[0]])
    assert collapse_dims([4, 1, 6], [[6, 6, 1]]) == ([24], [[1]])


@pytest.mark.parametrize("index", [
    (slice(1, 5),), (slice(None, None, -1),), (2,), (slice(1, None), slice(None, None, 2)),
    (-3, slice(None)), (slice(-4, None), 1), (slice(None, None, -2), slice(None, None, -1)),
    (slice(5, 2),), (slice(7, None, -3), slice(1, -1)),
])
def test_view_from_index_matches_numpy(index):
    a = np.arange(9 * 11).reshape(9, 11)
    shape, strides, off = hostops.view_from_index(a.shape, (11, 1), 0, index)
    got = np.lib.stride_tricks.as_strided(a.reshape(-1)[off:] if np.prod(shape) else a.reshape(-1),
                                          shape, [s * a.itemsize for s in strides])
    np.testing.assert_array_equal(got, a[index])


def test_view_from_index_errors():
    with pytest.raises(IndexError):
        hostops.view_from_index((3, 4), (4, 1), 0, (3,))
    with pytest.raises(IndexError):
        hostops.view_from_index((3, 4), (4, 1), 0, (0, 0, 0))


def test_host_scalar_eval_python_int_semantics():
    s = {"n_in": 2, "nodes": [{"op": "int_div", "in": [["i", 0], ["i", 1]], "dtype": "int64"},
                              {"op": "mod", "in": [["i", 0], ["i", 1]], "dtype": "int64"}],
         "out": [["t", 0], ["t", 1]]}
    q, r = hostops.eval_scalar_host(s, [np.int64(-7), np.int64(2)])
    assert (int(q), int(r)) == (-4, 1)


def test_codegen_sources_are_deterministic_and_keyed():
    sc = case_plan(_case("cfg2_gauss_sum")).nodes[3].params["scalar"]
    a = cg.KernelSpec(sc, ["float64"] * 4, ["float64"], [0], ["b", "c", "b", "b", "c"], 1, 4)
    b = cg.KernelSpec(sc, ["float64"] * 4, ["float64"], [0], ["b", "c", "b", "b", "c"], 1, 2)
    assert a.key() != b.key()
    assert cg.generate(a) == cg.generate(a)
    src, names = cg.generate(a)
    assert names[0] in src and "Pack<double, 4>" in src


@pytest.mark.parametrize("c", CASES, ids=CASE_IDS)
def test_dry_run_shapes_and_kernel_generation(c):
    """Full host path (fusion, broadcasting, views, dim collapsing, kernel specs, hiprtc
    cross-compilation of every generated kernel) with launches recorded instead of executed;
    output shapes/dtypes must equal the reference's."""
    plan = case_plan(c)
    ex = PlanExecutor(plan, dry_run=True)
    try:
        outs = ex(*case_inputs(c))
    except RuntimeError as e:
        # an extent that is COMPUTED on the device (Repeat with a vector of repeats: the output
        # length is the sum of the repeats) has no value in a dry run: everything up to that read
        # has been checked, the rest runs on the GPU (tests/test_gpu_parity.py)
        if "dry run: value of a computed device array is unknown" in str(e):
            assert any(n.op == "Searchsorted" for n in plan.nodes), c["name"]
            return
        raise
    # a do-while Scan's trip count is data dependent: the dry run (no values) runs n_steps
    # ... and Nonzero / boolean masks have value-dependent lengths (the dry run takes the maximum)
    data_dependent = any((n.op == "Scan" and n.params.get("as_while")) or n.op == "Nonzero" or
                         any("mask" in e for e in n.params.get("index", ())) for n in plan.nodes)
    for o, e in zip(outs, case_expected(c)):
        shape = o.shape if isinstance(o, DevArray) else np.shape(o)
        dtype = o.dtype if isinstance(o, DevArray) else np.asarray(o).dtype.name
        assert dtype == e.dtype.name and len(shape) == e.ndim
        assert data_dependent or tuple(shape) == e.shape
    assert all(name.startswith("ahip_") for name in ex.trace)


def test_elemwise_shape_mismatch_raises_reference_error():
    """reference: tensor/elemwise.py:733-735 'Shapes on dimension d do not match'."""
    c = _case("ew_bcast0_float64")
    ex = PlanExecutor(case_plan(c), dry_run=True)
    with pytest.raises(ValueError, match="Shapes on dimension 1 do not match"):
        ex(np.zeros((3, 5)), np.zeros((3, 4)))


def test_gemm_shape_errors():
    c = _case("gemm0_float64")
    ex = PlanExecutor(case_plan(c), dry_run=True)
    with pytest.raises(ValueError):
        ex(np.zeros((3, 5)), np.zeros((3, 4)), np.zeros((5, 5)))


def test_input_dtype_is_checked():
    c = _case("cfg1b_matrix_add")
    ex = PlanExecutor(case_plan(c), dry_run=True)
    with pytest.raises(TypeError):
        ex(np.zeros((3, 5), "float32"), np.zeros((3, 5)))


def test_subtensor_out_of_bounds_index_error():
    c = _case("subtensor_basic")
    ex = PlanExecutor(case_plan(c), dry_run=True)
    x = np.zeros((9, 11), "int32")
    with pytest.raises(IndexError):
        ex(x, np.int64(9))


def test_ifelse_runs_only_the_selected_branch():
    """IfElse is lazy (reference: ifelse.py:240 make_thunk + the VM's lazy scheduling): steps
    that only feed the branch not taken are never launched; nested IfElse keep their own."""
    c0, c1 = _case("ifelse_lazy_c0"), _case("ifelse_lazy_c1")
    ex = PlanExecutor(case_plan(c0), dry_run=True)
    assert ex._owned, "no step was found to be branch-exclusive"
    ex(*case_inputs(c0))
    t0 = list(ex.trace)
    ex.trace.clear()
    ex(*case_inputs(c1))
    t1 = list(ex.trace)
    # the else-branch of the first output is outer(y, y) @ y - 1: Ger + Gemv chain
    assert any(n in t0 for n in ("ahip_ger", "ahip_gemv", "ahip_gemv_epilogue"))
    assert not any(n in t1 for n in ("ahip_ger", "ahip_gemv", "ahip_gemv_epilogue"))


def _alias_plan(view_is_output):
    """z = alloc(v, n, 3); zT = z.T; r = set_subtensor(z[0], y) — the update is the last reader
    of z itself, but zT is a VIEW of the same buffer (executor.py _own_or_copy)."""
    from aesara_amd.plan import Node
    p = Plan("alias", {}, [], [], [])
    v = p.new_var("float64", [], "v")
    n = p.new_var("int64", [], "n")
    y = p.new_var("float64", [None], "y")
    c3 = p.add_const(np.int64(3))
    c0 = p.add_const(np.int64(0))
    z = p.new_var("float64", [None, None])
    zt = p.new_var("float64", [None, None])
    r = p.new_var("float64", [None, None])
    s = p.new_var("float64", [None])
    p.inputs = [v, n, y]
    p.nodes = [Node("Alloc", [v, n, c3], [z], {}),
               Node("DimShuffle", [z], [zt], {"new_order": [1, 0]})]
    upd = Node("IncSubtensor", [z, y, c0], [r], {"idx_list": [{"index": "in"}],
                                                  "set_instead_of_inc": True, "inplace": False})
    red = Node("CAReduce", [zt], [s], {"scalar_op": "add", "axis": [0], "acc_dtype": "float64"})
    if view_is_output:
        p.nodes += [upd]
        p.outputs = [zt, r]
    else:
        p.nodes += [upd, red]          # the view is READ after the update
        p.outputs = [s, r]
    return p


@pytest.mark.parametrize("view_is_output", [True, False])
def test_update_of_private_alloc_copies_when_a_view_is_still_live(view_is_output):
    ex = PlanExecutor(_alias_plan(view_is_output), dry_run=True)
    ex(np.float64(1.5), np.int64(4), np.arange(3.0))
    # the IncSubtensor must not have written into the Alloc buffer: a copy precedes it
    assert ex.trace.count("ahip_copy_strided") + ex.trace.count("ahip_copy") >= 2 or \
        any("copy" in t for t in ex.trace), ex.trace
    # without a live view the private buffer is still updated in place (no extra copy)
    p = _alias_plan(True)
    p.outputs = [p.outputs[1]]
    p.nodes = [n for n in p.nodes if n.op != "DimShuffle"]
    ex2 = PlanExecutor(p, dry_run=True)
    ex2(np.float64(1.5), np.int64(4), np.arange(3.0))
    assert len([t for t in ex2.trace if "copy" in t]) < len([t for t in ex.trace if "copy" in t])


def test_device_cell_checks_without_the_reference():
    import torch
    from aesara_amd.devcell import DeviceCell, PlainType
    cell = DeviceCell(PlainType("float32", (None, 3)), name="w", device=torch.device("cpu"))
    cell.value = np.ones((2, 3))                       # host value: filtered (cast) + "uploaded"
    assert isinstance(cell.storage[0], torch.Tensor) and cell.storage[0].dtype == torch.float32
    t = torch.zeros(5, 3)
    cell.value = t
    assert cell.storage[0] is t                        # device values are kept as they are
    for bad in (torch.zeros(5, 4), torch.zeros(5, 3, dtype=torch.float64), torch.zeros(3)):
        with pytest.raises(TypeError):
            cell.value = bad
    ro = DeviceCell(PlainType("float32", (None,)), readonly=True, storage=[None])
    with pytest.raises(Exception):
        ro.value = torch.zeros(3)


def test_device_filter_type_decisions():
    """``devcell.DeviceFilterType`` / ``filter_device_value``: the decisions of
    ``TensorType.filter`` (reference tensor/type.py:135-256) for a value that is already a tensor
    — strict dtype, refused downcast, rank, static shape — and delegation of everything else to
    the wrapped type (CPU tensors stand in for device tensors: no conversion kernel is reached)."""
    import copy
    import pickle
    import torch
    from aesara_amd.devcell import DeviceFilterType, PlainType, filter_device_value
    base = PlainType("float32", (None, 3))
    t = DeviceFilterType(base)
    ok = torch.zeros(5, 3)
    assert filter_device_value(base, ok) is ok and filter_device_value(base, ok, strict=True) is ok
    with pytest.raises(TypeError, match="expected a tensor with dtype=float32"):
        filter_device_value(base, ok.double(), strict=True)
    with pytest.raises(TypeError, match="without risking loss of precision"):
        filter_device_value(base, ok.double())                 # float64 -> float32 is a downcast
    with pytest.raises(TypeError, match="Wrong number of dimensions"):
        filter_device_value(base, torch.zeros(3))
    with pytest.raises(TypeError, match="not compatible"):
        filter_device_value(base, torch.zeros(5, 4))
    with pytest.raises(TypeError, match="Wrong number of dimensions"):
        filter_device_value(base, torch.zeros(3, dtype=torch.int8))   # rank is checked before a cast
    # a CPU tensor goes to the wrapped type's host filter (np.asarray works on it), as do arrays
    out = t.filter(torch.ones(2, 3, dtype=torch.float64), allow_downcast=True)
    assert isinstance(out, np.ndarray) and out.dtype == np.float32
    assert t.filter(np.ones((2, 3), "float32")).shape == (2, 3)
    # the wrapper IS the wrapped type for everything but filter
    assert t == base and base == t._t and hash(t) == hash(base) and repr(t) == repr(base)
    assert t.dtype == "float32" and t.ndim == 2 and t.shape == (None, 3)
    assert DeviceFilterType(t)._t is base                     # never nested
    t2 = copy.deepcopy(t)
    assert isinstance(t2, DeviceFilterType) and t2._t is not base and t2.dtype == "float32"
    t3 = pickle.loads(pickle.dumps(t))
    assert isinstance(t3, DeviceFilterType) and t3.shape == (None, 3)


@pytest.mark.parametrize("name", ["scan_embedding_lookup_in_step", "scan_embedding_lookup_batch_f32"])
def test_index_glue_on_sequence_rows_leaves_the_step(name):
    """fusion.push_out_sequence_glue: ``E[idx_t]`` (scalar index / index vector of a batch) and the
    ``set_subtensor(zeros[k_t], 1)`` unit vectors of a recurrent step are restated over whole
    sequences in front of the Scan and enter it as sequences; the rewritten plan gives the
    REFERENCE's outputs (oracle), the forward step has no index node left, and the dry run puts the
    forward loop on a persistent kernel."""
    import interp
    from golden_util import CASES, assert_matches, case_expected, case_inputs, case_plan
    from aesara_amd.executor import PlanExecutor
    from aesara_amd.fusion import push_out_sequence_glue
    c = next(c for c in CASES if c["name"] == name)
    plan = case_plan(c)
    p2 = push_out_sequence_glue(plan)
    assert p2 is not plan
    fwd = next(n for n in p2.nodes if n.op == "Scan")
    assert not {"Subtensor", "AdvancedSubtensor1", "IncSubtensor", "ScalarFromTensor"} & {n.op for n in fwd.params["inner"].nodes}
    assert fwd.params["n_seqs"] > next(n for n in plan.nodes if n.op == "Scan").params["n_seqs"]
    assert len(fwd.params["inner"].inputs) == len(fwd.inputs) - 1 - fwd.params["n_nit_sot"] + \
        sum(len(t) - 1 for t in fwd.params["mit_sot_in_slices"] + fwd.params.get("mit_mot_in_slices", []))
    for g, e in zip(interp.run_plan(p2, case_inputs(c)), case_expected(c)):
        assert_matches(c, g, e)
    assert push_out_sequence_glue(p2) is p2              # nothing left to take out
    # ... and the gradient Scan's ``dE[idx_t] += delta_t`` (a recurrent scatter into the table's
    # gradient) leaves ITS loop as one scatter-add behind it (push_out_product_accumulators, the
    # two-row buffer form: [acc_{T-1}, acc_T]); the plan with every Scan in that form still gives
    # the reference's outputs
    from aesara_amd.fusion import push_out_product_accumulators, read_last_row_only, zero_filled_vars
    from aesara_amd.plan import Node, Plan
    zeros, last = zero_filled_vars(p2), read_last_row_only(p2)
    nodes, ren, vars_, sunk = [], {}, dict(p2.vars), 0
    for node in p2.nodes:
        node = Node(node.op, [ren.get(i, i) for i in node.inputs], list(node.outputs), node.params)
        if node.op == "Scan":
            need, same, le2 = [], [], []
            wrap = Plan("w", vars_, list(node.inputs), list(node.outputs), [node])
            w2 = push_out_product_accumulators(wrap, need, same, zeros=zeros, last_only=last, need_le2=le2)
            if w2 is not wrap:
                assert le2 and not need and not same
                sunk += 1
                vars_ = w2.vars
                nodes.extend(w2.nodes)
                ren.update({o: o2 for o, o2 in zip(node.outputs, w2.outputs) if o != o2})
                grad_scan = next(n for n in w2.nodes if n.op == "Scan")
                assert not {"IncSubtensor", "AdvancedIncSubtensor1", "ScalarFromTensor"} & \
                    {n.op for n in grad_scan.params["inner"].nodes}
                continue
        nodes.append(node)
    assert sunk == 1
    p3 = Plan(p2.name, vars_, list(p2.inputs), [ren.get(o, o) for o in p2.outputs], nodes)
    for g, e in zip(interp.run_plan(p3, case_inputs(c)), case_expected(c)):
        assert_matches(c, g, e)
    ex = PlanExecutor(plan, dry_run=True)
    ex(*case_inputs(c))
    assert set(ex.scan_modes.values()) == {"persistent"} and len(ex.scan_modes) == 2, ex.scan_modes


@pytest.mark.parametrize("name", ["scan_nitsot_map", "scan_map_jacobian_rows", "scan_map_hessian_unit_vectors",
                                  "scan_map_rows_reduce_broadcast"])
def test_scans_without_recurrence_restated_over_whole_sequences(name):
    """fusion.batch_map_step: every Scan without recurrence of these goldens is replaced by ONE
    evaluation of the restated step plan over whole sequences (host logic of
    ``ScanMixin._scan_all_rows`` restated in NumPy: first n_steps rows in, last ``keep`` rows out) —
    the outputs must still be the REFERENCE's (tests/golden), all through the oracle."""
    import copy
    import interp
    from golden_util import CASES, assert_matches, case_expected, case_inputs, case_plan
    from aesara_amd.fusion import batch_map_step
    from aesara_amd.plan import Node
    c = next(c for c in CASES if c["name"] == name)
    plan = copy.copy(case_plan(c))
    nodes, swapped = [], 0
    for node in plan.nodes:
        p = node.params
        if node.op != "Scan" or p.get("as_while") or p.get("n_shared_outs", 0) or p.get("mit_mot_in_slices") \
                or p["mit_sot_in_slices"] or p["sit_sot_in_slices"]:
            nodes.append(node)
            continue
        r = batch_map_step(p["inner"], p["n_seqs"])
        assert r is not None, [n.op for n in p["inner"].nodes]

        def all_rows(*a, bp=r["plan"], n_seqs=p["n_seqs"], n_nit=p["n_nit_sot"]):
            n_steps = int(a[0])
            seqs = [np.asarray(s)[:n_steps] for s in a[1:1 + n_seqs]]
            keep = [int(k) for k in a[1 + n_seqs:1 + n_seqs + n_nit]]
            rows = interp.run_plan(bp, seqs + list(a[1 + n_seqs + n_nit:]))
            assert all(x.shape[0] == n_steps for x in rows)
            return [x[n_steps - k:] for x, k in zip(rows, keep)]
        nodes.append(Node("HostCall", list(node.inputs), list(node.outputs), {"fn": all_rows}))
        swapped += 1
    assert swapped
    plan.nodes = nodes
    got = interp.run_plan(plan, case_inputs(c))
    for g, e in zip(got, case_expected(c)):
        assert_matches(c, g, e)


@pytest.mark.parametrize("name", ["gru_bptt_b1_f32", "gru_bptt_b4_f32", "lstm_bptt_vec_f32", "cfg4_gru_b1_f32"])
def test_sequence_only_hoisting_preserves_the_step(name):
    """fusion.hoist_sequence_only on the Scan inner plans of the golden recurrences and their
    gradient Scans: the lifted plan evaluated ONCE over whole sequences plus the reduced step plan
    must give, for every step, what the original step plan gives (all three through the oracle)."""
    import interp
    from golden_util import CASES, case_plan
    from aesara_amd.fusion import hoist_sequence_only, split_invariant
    plan = case_plan(next(c for c in CASES if c["name"] == name))
    rng = np.random.default_rng(5)
    T, B, H = 6, 3, 8
    checked = 0
    for node in plan.nodes:
        if node.op != "Scan":
            continue
        p, inner = node.params, node.params["inner"]
        n_seqs = p["n_seqs"]
        n_var = len(inner.inputs) - p["n_non_seqs"]
        pre, loop, hoisted = split_invariant(inner, inner.inputs[n_var:])
        loop2, lifted = hoist_sequence_only(loop, list(loop.inputs[:n_seqs]), set(loop.inputs[n_var:]))
        if lifted is None:
            continue

        def rand(v, lead=()):
            shp = tuple((1 if s == 1 else (B if k == 0 and inner.vars[v].ndim == 2 and name.endswith("b4_f32") else H))
                        for k, s in enumerate(inner.vars[v].shape))
            return rng.standard_normal(lead + shp).astype(inner.vars[v].dtype)
        seqs = [rand(v, (T,)) for v in inner.inputs[:n_seqs]]
        taps = [rand(v) for v in inner.inputs[n_seqs:n_var]]
        invs = [rand(v) for v in inner.inputs[n_var:]]
        if name.endswith("b4_f32"):          # invariant matrices are H x H, not B x H
            invs = [rng.standard_normal((H, H)).astype("float32") if a.ndim == 2 else a for a in invs]
        hv = interp.run_plan(pre, invs) if pre is not None else []
        inv_all = dict(zip(list(inner.inputs[n_var:]) + list(hoisted), invs + list(hv)))
        stacked = lifted.get("stacked", False)
        largs = []
        for v in lifted["seq_in"]:
            S = seqs[list(inner.inputs[:n_seqs]).index(v)]
            largs.append(S.reshape(T * S.shape[1], S.shape[2]) if stacked else S)
        largs += [inv_all[v] for v in lifted["inv_in"]]
        rows = interp.run_plan(lifted["plan"], largs)
        if stacked:
            rows = [r.reshape(T, -1, r.shape[1]) for r in rows]
        for t in range(T):
            step_in = [s[t] for s in seqs] + taps + invs
            want = interp.run_plan(inner, step_in)
            got = interp.run_plan(loop2, step_in + list(hv) + [r[t] for r in rows])
            for g, w in zip(got, want):
                np.testing.assert_allclose(g, w, rtol=2e-5, atol=2e-6)
        checked += 1
    assert checked >= 1


@pytest.mark.parametrize("name", ["lstm_bptt_float32", "lstm_fused_fwd_f32", "lstm_bptt_float64",
                                  "lstm_fused_vec_f32"])
def test_column_slice_splitting_preserves_the_step(name):
    """fusion.split_column_slices on fused-gate LSTM steps (forward and gradient inner plans): the
    rewritten step (one product chain per gate, column views of the invariant operands) gives the
    same outputs as the original one, through the oracle; the forward step must actually change."""
    import interp
    from golden_util import CASES, case_plan
    from aesara_amd.fusion import split_column_slices
    plan = case_plan(next(c for c in CASES if c["name"] == name))
    changed = 0
    for node in plan.nodes:
        if node.op == "Scan":
            inner, p = node.params["inner"], node.params
            n_var = len(inner.inputs) - p["n_non_seqs"]
            changed += split_column_slices(inner, set(inner.inputs[n_var:])) is not inner
    assert changed >= 1
    # end-to-end through the oracle: the whole plan with every Scan inner plan rewritten
    import copy
    from golden_util import case_inputs
    c = next(c for c in CASES if c["name"] == name)
    ins = case_inputs(c)
    want = interp.run_plan(plan, ins)
    plan2 = copy.deepcopy(plan)
    for node in plan2.nodes:
        if node.op == "Scan":
            inner = node.params["inner"]
            n_var = len(inner.inputs) - node.params["n_non_seqs"]
            node.params["inner"] = split_column_slices(inner, set(inner.inputs[n_var:]))
    got = interp.run_plan(plan2, ins)
    for g, w in zip(got, want):
        np.testing.assert_allclose(g, w, rtol=1e-5 if "32" in name else 1e-12, atol=1e-6 if "32" in name else 1e-13)


def test_accumulator_push_out_preserves_every_golden_scan_plan():
    """fusion.push_out_accumulators on every golden plan with a Scan: where a sit-sot accumulator
    (bias gradients of batched recurrences, summed inside the gradient Scan) is rebuilt after the
    loop, the rewritten plan gives the same outputs through the oracle — whatever row of the
    accumulator buffer the outer graph reads."""
    import interp
    from golden_util import CASES, case_inputs, case_plan
    from aesara_amd.fusion import push_out_accumulators
    changed = []
    for c in CASES:
        plan = case_plan(c)
        if not any(n.op == "Scan" for n in plan.nodes):
            continue
        new = push_out_accumulators(plan)
        if new is plan:
            continue
        changed.append(c["name"])
        ins = case_inputs(c)
        for g, w in zip(interp.run_plan(new, ins), interp.run_plan(plan, ins)):
            np.testing.assert_allclose(np.asarray(g, dtype="float64"), np.asarray(w, dtype="float64"),
                                       rtol=1e-6, atol=1e-7, err_msg=c["name"])
    assert {"rnn_bias_bptt_b4_f32", "lstm_fused_bptt_h64_f32"} <= set(changed), changed


@pytest.mark.parametrize("name", ["lstm_fused_bptt_h64_f32", "lstm_bptt_float32", "lstm_bptt_float64",
                                  "lstm_fused_vec_bptt_f32"])
def test_assembled_gate_gradients_leave_the_step(name):
    """fusion.split_assembled_columns (after push_out_accumulators) on fused-gate LSTM gradient
    Scans: the 4H-wide assembled gate gradient becomes per-gate nit-sot outputs + one Join after the
    loop, its product a sum over row blocks of the weight matrix — same outputs through the oracle."""
    import interp
    from golden_util import CASES, case_inputs, case_plan
    from aesara_amd.fusion import push_out_accumulators, split_assembled_columns
    c = next(c for c in CASES if c["name"] == name)
    plan, ins = case_plan(c), case_inputs(c)
    p1 = push_out_accumulators(plan)
    p2 = split_assembled_columns(p1)
    assert p2 is not p1
    scans = [n for n in p2.nodes if n.op == "Scan"]
    assert scans[1].params["n_nit_sot"] == 4 and not any(
        m.op in ("IncSubtensor", "Alloc", "Assert", "CAReduce") for m in scans[1].params["inner"].nodes)
    tol = 1e-6 if "32" in name else 1e-13
    for g, w in zip(interp.run_plan(p2, ins), interp.run_plan(plan, ins)):
        np.testing.assert_allclose(np.asarray(g, "float64"), np.asarray(w, "float64"), rtol=tol, atol=tol)


def test_column_slice_splitting_keeps_broadcast_operands_whole():
    """an invariant scalar handed in as [1, 1] (``DimShuffle('x','x')`` of a shared
    scalar) multiplies the fused product before it is sliced per gate.  The rewritten step must
    leave that operand unsliced (it has no columns a:b) — it used to be cut to an empty [1, 0]
    array for every gate with a >= 1."""
    import interp
    from aesara_amd.fusion import split_column_slices
    from aesara_amd.plan import Node, Plan
    p = Plan("bcast_gate", {}, [], [], [])
    h = p.new_var("float64", [None, None], "h")          # [B, H]
    W = p.new_var("float64", [None, None], "W")          # [H, 2H] invariant
    s = p.new_var("float64", [1, 1], "s")                # invariant scalar as [1, 1]
    b = p.new_var("float64", [1, None], "b")             # invariant row [1, 2H]
    p.inputs = [h, W, s, b]
    prod = p.new_var("float64", [None, None])
    p.nodes.append(Node("Dot22", [h, W], [prod], {}))
    mul = {"n_in": 3, "nodes": [{"op": "mul", "in": [["i", 0], ["i", 1]], "dtype": "float64"},
                                {"op": "add", "in": [["t", 0], ["i", 2]], "dtype": "float64"}],
           "out": [["t", 1]]}
    pre = p.new_var("float64", [None, None])
    p.nodes.append(Node("Elemwise", [prod, s, b], [pre], {"scalar": mul}))
    H = 5
    outs = []
    for k in range(2):
        o = p.new_var("float64", [None, None])
        p.nodes.append(Node("Subtensor", [pre], [o],
                            {"idx_list": [{"slice": [None, None, None]}, {"slice": [k * H, (k + 1) * H, None]}]}))
        t = p.new_var("float64", [None, None])
        p.nodes.append(Node("Elemwise", [o], [t], {"scalar": {
            "n_in": 1, "nodes": [{"op": "tanh", "in": [["i", 0]], "dtype": "float64"}], "out": [["t", 0]]}}))
        outs.append(t)
    p.outputs = outs
    rng = np.random.default_rng(0)
    ins = [rng.standard_normal((3, H)), rng.standard_normal((H, 2 * H)), np.array([[0.7]]),
           rng.standard_normal((1, 2 * H))]
    want = interp.run_plan(p, ins)
    q = split_column_slices(p, {W, s, b})
    assert q is not p and sum(n.op == "Dot22" for n in q.nodes) == 2     # one product per gate
    got = interp.run_plan(q, ins)
    for g, w in zip(got, want):
        np.testing.assert_allclose(g, w, rtol=1e-13)
    # a chain VALUE that is broadcast along the sliced axis is outside the pattern: left alone
    p2 = Plan("bcast_chain", dict(p.vars), list(p.inputs), list(p.outputs), list(p.nodes))
    p2.vars = {k: type(v)(v.id, v.dtype, list(v.shape), v.name, v.const) for k, v in p.vars.items()}
    p2.vars[pre].shape = [None, 1]
    assert split_column_slices(p2, {W, s, b}) is p2


def test_accumulator_push_out_with_a_buffer_longer_than_the_loop():
    """when the caller's accumulator buffer holds MORE than T + 1 rows (truncated
    BPTT) the reference leaves the rows beyond T zero (scan/op.py:2139-2144); the rebuilt buffer
    must do the same instead of slicing from a negative start."""
    import copy
    import interp
    from golden_util import CASES, case_inputs, case_plan
    from aesara_amd.fusion import push_out_accumulators
    from aesara_amd.plan import Node
    c = next(c for c in CASES if c["name"] == "rnn_bias_bptt_b4_f32")
    plan = copy.deepcopy(case_plan(c))
    ref = push_out_accumulators(case_plan(c))
    # which Scan input is the accumulator that leaves the loop: the one the rewrite reads rows of
    scan_i = next(i for i, n in enumerate(plan.nodes) if n.op == "Scan" and n.params.get("mit_mot_in_slices"))
    scan = plan.nodes[scan_i]
    p = scan.params
    n_mm, n_ms = len(p["mit_mot_in_slices"]), len(p["mit_sot_in_slices"])
    base_in = 1 + p["n_seqs"] + n_mm + n_ms
    new_scan = next(n for n in ref.nodes if n.op == "Scan" and n.params.get("mit_mot_in_slices"))
    q = next(k for k in range(len(p["sit_sot_in_slices"]))
             if scan.inputs[base_in + k] not in new_scan.inputs)
    init_buf = scan.inputs[base_in + q]
    acc_out = scan.outputs[n_mm + n_ms + q]
    v = plan.vars[init_buf]
    # init' = [init; 20 rows of garbage] (more than T + 1 rows): the Scan zeroes rows T+1.. itself
    w = plan.new_var("int64", [])
    padb = plan.new_var(v.dtype, [None] + list(v.shape[1:]))
    three = plan.add_const(20, "int64")
    init2 = plan.new_var(v.dtype, [None] + list(v.shape[1:]))
    pre = [Node("Shape_i", [init_buf], [w], {"i": 1}),
           Node("Alloc", [plan.add_const(7.0, v.dtype), three, w], [padb], {}),
           Node("Join", [plan.add_const(0, "int8"), init_buf, padb], [init2], {})]
    scan.inputs[base_in + q] = init2
    plan.nodes[scan_i:scan_i] = pre
    plan.outputs = list(plan.outputs) + [acc_out]           # the whole buffer, tail included
    ins = case_inputs(c)
    want = interp.run_plan(plan, ins)
    new = push_out_accumulators(plan)
    assert new is not plan
    got = interp.run_plan(new, ins)
    assert want[-1].shape == got[-1].shape and np.all(want[-1][-3:] == 0) and np.any(want[-1][:-12] != 0)
    for g, w_ in zip(got, want):
        np.testing.assert_allclose(np.asarray(g, "float64"), np.asarray(w_, "float64"), rtol=1e-6, atol=1e-7)


def test_shared_left_dots_become_one_wide_product():
    """fusion.merge_shared_left_dots: three ``Dot22(x, W_k)`` of one operand -> one product against
    the joined weights + column views; same values through the oracle, other nodes untouched."""
    import interp
    from aesara_amd.fusion import merge_shared_left_dots
    from aesara_amd.plan import Node, Plan
    p = Plan("gates", {}, [], [], [])
    x = p.new_var("float64", [None, None], "x")
    Ws = [p.new_var("float64", [None, None], "W%d" % k) for k in range(3)]
    y = p.new_var("float64", [None, None], "y")
    p.inputs = [x] + Ws + [y]
    outs = []
    for W in Ws:
        o = p.new_var("float64", [None, None])
        p.nodes.append(Node("Dot22", [x, W], [o], {}))
        outs.append(o)
    lone = p.new_var("float64", [None, None])
    p.nodes.append(Node("Dot22", [y, Ws[0]], [lone], {}))          # another left operand: kept
    p.outputs = outs + [lone]
    rng = np.random.default_rng(0)
    ins = [rng.standard_normal((7, 5))] + [rng.standard_normal((5, n)) for n in (4, 6, 3)] + \
        [rng.standard_normal((2, 5))]
    q = merge_shared_left_dots(p)
    assert q is not p and sum(n.op == "Dot22" for n in q.nodes) == 2 and any(n.op == "Join" for n in q.nodes)
    for g, w in zip(interp.run_plan(q, ins), interp.run_plan(p, ins)):
        np.testing.assert_allclose(g, w, rtol=1e-13)
    single = Plan("one", dict(p.vars), list(p.inputs), [lone], [p.nodes[3]])
    assert merge_shared_left_dots(single) is single


def test_sequence_product_windows():
    """scan_persist_mat.xfold_windows: which window of the step each in-loop sequence product runs
    in.  GRU (phase 0: r on h; phase 1: z on h — already in registers — and the candidate on r*h):
    the products of phase 1 run in front of phase 0's fetch, phase 0's own in the LAST window of
    the step before (from x_{t+1}); a recurrence with one fetching phase runs a whole step ahead;
    a product that would put x_t and x_{t+1} into one window is refused."""
    from types import SimpleNamespace
    from aesara_amd import scan_persist_mat as sm
    gru = SimpleNamespace(state={1: 0}, phases=[{"dots": [(5, 1)]}, {"dots": [(3, 1), (7, 11)]}])
    assert [(len(e), len(l), f) for e, l, f in sm.phase_fetches(gru)] == \
        [(0, 1, [(1, "prev")]), (1, 1, [(11, "cur")])]
    w = sm.xfold_windows(gru, [("r", 0, 0), ("h", 1, 1), ("z", 1, 0)])
    assert w == {"win": {0: [1, 2], 1: [0]}, "wrapped": {0}, "reload": 0, "ahead": 1}
    # only phase-0 products folded: everything a step ahead, x moves on behind the last window
    assert sm.xfold_windows(gru, [("r", 0, 0)]) == {"win": {0: [], 1: [0]}, "wrapped": {0}, "reload": 1, "ahead": 2}
    elman = SimpleNamespace(state={1: 0}, phases=[{"dots": [(2, 1)]}])
    assert sm.xfold_windows(elman, [("x", 0, 0)]) == {"win": {0: [0]}, "wrapped": {0}, "reload": 0, "ahead": 2}
    # phase 2 (no fetch of its own) would share the window of phase 1 with phase 0's wrapped product
    three = SimpleNamespace(state={1: 0}, phases=[{"dots": [(5, 1)]}, {"dots": [(7, 11)]}, {"dots": [(3, 1)]}])
    assert sm.xfold_windows(three, [("a", 0, 0), ("c", 2, 0)]) is None
    assert sm.xfold_windows(gru, []) is None


@pytest.mark.parametrize("name,folded", [("xfold_gru_b16_f32", 3), ("xfold_rnn_b16_f32", 1)])
def test_sequence_products_move_into_the_loop_at_eligible_shapes(name, folded, monkeypatch):
    """Dry run of the two golden recurrences with batch % 16 == 0 and state % 64 == 0: the Scan
    takes the persistent matrix kernel with the sequence products inside (every weight block fits
    in LDS at H = 64; the kernel is generated and compiled on the way) and AESARA_HIP_SM_XFOLD=0
    keeps them up front."""
    from aesara_amd import scan_persist_mat as sm
    c = _case(name)
    seen = []

    class Spy(sm.SpecMat):            # (a cached kernel is not generated again: watch the specs)
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            seen.append(self)
    monkeypatch.setattr(sm, "SpecMat", Spy)
    ex = PlanExecutor(case_plan(c), dry_run=True)
    ex(*case_inputs(c))
    assert list(ex.scan_modes.values()) == ["persistent"]
    assert list(ex.scan_notes.values()) == ["sequence products in the loop"]
    xf = [s for s in seen if s.xfold]
    assert xf and len(xf[-1].xfold["items"]) == folded and xf[-1].xmode == "frag"
    monkeypatch.setenv("AESARA_HIP_SM_XFOLD", "0")
    ex = PlanExecutor(case_plan(c), dry_run=True)
    ex(*case_inputs(c))
    assert list(ex.scan_modes.values()) == ["persistent"] and not ex.scan_notes


_PRODUCT_ACC = ["sp_taps_bptt_f64", "sm_taps_bptt_b16_f32", "scan_variant_1", "scan_variant_2", "scan_variant_4",
                "scan_variant_5", "scan_variant_7", "scan_variant_8", "scan_variant_10", "scan_variant_11"]


@pytest.mark.parametrize("name", _PRODUCT_ACC)
def test_product_accumulators_leave_the_loop_as_one_product(name):
    """fusion.push_out_product_accumulators (what PushOutDot1 of the reference, scan/rewriting.py,
    does for the one-tap gradient only): the weight gradient a gradient Scan sums inside its loop —
    a sit-sot ``Gemm(acc, 1, A_t, B_t, 1)`` / ``Ger(acc, 1, x_t, y_t)`` — becomes one product over
    the stacked per-step operands behind the loop.  The rewritten plan, run by the NumPy oracle, must
    reproduce the REFERENCE's outputs; the Scan must have lost the accumulator (and nothing else);
    the runtime conditions (one-row buffer, nit-sot outputs that keep every row) are reported."""
    import interp
    from golden_util import assert_matches
    from aesara_amd.fusion import push_out_accumulators, push_out_product_accumulators
    c = next(c for c in CASES if c["name"] == name)
    p1 = push_out_accumulators(case_plan(c))
    # (without the lists only buffers whose length is STATICALLY one row are rewritten: an Alloc of
    # a shape expression usually is not — the executor decides per call)
    one_row, equal = [], []
    p2 = push_out_product_accumulators(p1, one_row, equal)
    assert p2 is not p1 and len(one_row) == 1
    s1 = [n for n in p1.nodes if n.op == "Scan"]
    s2 = [n for n in p2.nodes if n.op == "Scan"]
    assert len(s1) == len(s2)
    changed = [(a, b) for a, b in zip(s1, s2)
               if len(a.params["sit_sot_in_slices"]) != len(b.params["sit_sot_in_slices"])]
    assert len(changed) == 1
    a, b = changed[0]
    assert len(a.params["sit_sot_in_slices"]) - len(b.params["sit_sot_in_slices"]) == 1
    assert len(b.inputs) <= len(a.inputs) and len(b.outputs) <= len(a.outputs)
    assert not any(n.op in ("Gemm", "Ger") and b.params["inner"].inputs and
                   n.inputs[0] in a.params["inner"].inputs and n.inputs[0] not in b.params["inner"].inputs
                   for n in b.params["inner"].nodes)
    got = interp.run_plan(p2, case_inputs(c))
    assert_matches(c, got, case_expected(c), "accumulator behind the loop")
    # and through the executor's own dispatch (dry run: kernels generated and compiled, no launch):
    # the two-tap cases without reductions in the forward step run both Scans persistent
    ex = PlanExecutor(case_plan(c), dry_run=True)
    ex(*case_inputs(c))
    if name in ("sp_taps_bptt_f64", "sm_taps_bptt_b16_f32", "scan_variant_1", "scan_variant_4",
                "scan_variant_7", "scan_variant_10"):
        assert list(ex.scan_modes.values()) == ["persistent", "persistent"], ex.scan_modes


def test_design_lists_every_switch_of_the_registry():
    """The switch table (``docs/SWITCHES.md``, DESIGN §3.5 points at it) is generated from
    ``aesara_amd/knobs.py`` (``knobs.table()``): every registered switch appears there with its
    current default, and nothing outside the registry reads ``AESARA_HIP_*`` from the environment."""
    import os
    import re
    from aesara_amd import knobs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    design = open(os.path.join(root, "docs", "SWITCHES.md")).read()
    for name, default, _cur, _doc in knobs.table():
        m = re.search(r"\| `%s` \| ([^|]*) \|" % re.escape(name), design)
        assert m, f"{name} missing from docs/SWITCHES.md"
        assert m.group(1).strip() == ("—" if default is None else str(default)), (name, m.group(1), default)
    pkg = os.path.join(root, "aesara_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py") and fn != "knobs.py":
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"environ[^\n]*AESARA_HIP_", src), f"{fn} reads an AESARA_HIP_* switch directly"


@pytest.mark.parametrize("B,launches,folded", [(64, 1, True), (128, 2, True), (200, 4, False), (256, 4, True)])
def test_large_batches_run_as_slices_of_the_batch(B, launches, folded):
    """A forward matrix-state Scan whose 16 x 16 tiles outnumber the CUs (dry run: 256) is handed to
    the one-block-per-workgroup kernel slice by slice (exec_scan._scan_persist_mat_chunks): one
    persistent launch per slice of 64 batch rows at H = 1024, the in-loop sequence products kept
    when every slice is a multiple of 16 rows."""
    plan = case_plan(next(c for c in CASES if c["name"] == "cfg4_gru_b8_f32"))
    T, H = 6, 1024
    Ws = [np.zeros((H, H), "float32") for _ in range(6)]
    ex = PlanExecutor(plan, dry_run=True)
    ex(np.zeros((T, B, H), "float32"), np.zeros((B, H), "float32"), *Ws)
    assert list(ex.scan_modes.values()) == ["persistent"], ex.scan_modes
    assert sum(1 for t in ex.trace if t == "ahip_launch_p") == launches
    assert bool(ex.scan_notes) == folded, ex.scan_notes


def test_two_tap_gradient_scan_maps_to_two_states_on_one_buffer():
    """scan_persist.analyze on the gradient Scan of a two-tap recurrence (after the weight-gradient
    product left the loop): mit-mot taps [0, 2, 1] -> [2, 1] = two states — tap 0 with the output
    slot of out-tap 1 (the group's own slot), tap 1 with a further slot on the same buffer for
    out-tap 2 — and tap 2 as the buffer-resident sequence."""
    from aesara_amd import scan_persist as sp
    from aesara_amd.fusion import push_out_product_accumulators
    c = next(c for c in CASES if c["name"] == "sm_taps_bptt_b16_f32")
    plan = push_out_product_accumulators(case_plan(c), [], [])
    node = [n for n in plan.nodes if n.op == "Scan"][1]
    p = node.params
    assert p["mit_mot_in_slices"] == [[0, 2, 1]] and p["mit_mot_out_slices"] == [[2, 1]] and not p["sit_sot_in_slices"]
    inner = PlanExecutor(p["inner"], dry_run=True)
    prog, why = sp.analyze(inner, p, 0)
    assert prog is not None, why
    ins = p["inner"].inputs
    t0, t2, t1 = ins[p["n_seqs"]:p["n_seqs"] + 3]
    n_outer = 1 + p["n_nit_sot"]
    assert prog.mm_extra == [(n_outer, 0, 2)]
    assert prog.state[t0] == 0 and prog.state[t1] == n_outer          # slots of out-taps 1 and 2
    assert prog.tap_seq == {t2: 0} and prog.tap_top == {t2: 2}
    outs = p["inner"].outputs                                         # [out-tap 2, out-tap 1, nit-sots ...]
    assert prog.new_of_state[t1] == outs[0] and prog.new_of_state[t0] == outs[1]


def test_streaming_policy_for_read_once_operands(monkeypatch):
    """exec_elemwise.BIG_STREAM: a flat stream / Gemv matrix with an operand of 96 MiB or more takes
    16 bytes per lane and load with non-temporal accesses (and the capped launch for plain
    Elemwise); smaller operands keep 32-byte packs and cached loads; setting AESARA_HIP_NT (or
    VECBYTES) switches the policy off.  Dry runs: which kernel specs and C-ABI entry points."""
    from aesara_amd import exec_common as ec
    from aesara_amd.device import DevArray, contiguous_strides
    from aesara_amd.executor import PlanExecutor
    from golden_util import CASES, case_plan

    def fake(shape, dtype):
        n = int(np.prod(shape))
        return DevArray(ec._FakeBuf(max(n, 1), dtype), 0, tuple(shape), contiguous_strides(shape), dtype)

    def case(name):
        return next(c for c in CASES if c["name"] == name)
    specs = []
    orig = ec._Kernels.get.__func__

    def get(cls, spec, load=True):
        specs.append(spec)
        return orig(cls, spec, load=load)
    monkeypatch.setattr(ec._Kernels, "get", classmethod(get))

    def run(name, args):
        specs.clear()
        ex = PlanExecutor(case_plan(case(name)), dry_run=True)
        ex(*args)
        return list(specs), list(ex.trace)
    big2 = [fake((4096, 4096), "float64"), np.float64(0.1), np.float64(1.3)]
    sp, _ = run("cfg2_gauss_sum", big2)
    assert [(s.vec, int(s.nt), s.unroll) for s in sp if s.reduce] == [(2, 3, 2)]
    sp, _ = run("cfg2_gauss_sum", [fake((2048, 2048), "float64"), np.float64(0.1), np.float64(1.3)])
    assert [(s.vec, int(s.nt)) for s in sp if s.reduce] == [(4, 0)]            # 32 MiB: cached 32-byte packs
    sp, tr = run("cfg1b_matrix_add", [fake((4096, 4096), "float64")] * 2)
    assert [(s.vec, int(s.nt)) for s in sp] == [(2, 3)] and "ahip_elemwise_wg" in tr and "ahip_elemwise" not in tr
    sp, tr = run("cfg1b_matrix_add", [fake((1024, 1024), "float64")] * 2)
    assert [(s.vec, int(s.nt)) for s in sp] == [(4, 0)] and "ahip_elemwise" in tr and "ahip_elemwise_wg" not in tr
    monkeypatch.setenv("AESARA_HIP_NT", "0")                                   # explicit: policy off
    sp, tr = run("cfg1b_matrix_add", [fake((4096, 4096), "float64")] * 2)
    assert [(s.vec, int(s.nt)) for s in sp] == [(4, 0)] and "ahip_elemwise" in tr


def test_which_golden_scans_run_as_one_launch():
    """The persistent-Scan class over the whole golden set, decided on the host (dry runs: analysis,
    layout checks and kernel generation, no device): of 105 Scans that keep a loop 95 take a one-launch kernel and 6
    (no recurrence) are one evaluation over whole sequences; the four that do neither are named with the reason ``PlanExecutor.scan_modes`` gives — a silent
    fall-back to the launch list of any other golden fails here, on CPU."""
    from aesara_amd.executor import PlanExecutor
    from golden_util import CASES, case_plan
    from golden_inputs import make_input
    expected_launch_list = {
        "rnn_lm_loss_and_grads": "weight layout",                                     # non-square weights of a matrix state
        "scan_nested_with_grad": "Shape_i",                                           # a shape node inside the step
        "scan_seq_products_two_row_counts": "Gemm",                                   # a bare Gemm node in the step
        "sp_rnn_proj_narrow_f32": "matrix layout",                                    # projection narrower than the state
    }
    total = persistent = all_rows = 0
    for c in CASES:
        if "Scan" not in json.dumps(c["plan"]["nodes"])[:200000] and not any(
                n["op"] == "Scan" for n in c["plan"]["nodes"]):
            continue
        ex = PlanExecutor(case_plan(c), dry_run=True)
        try:
            ex(*[make_input(s) for s in c["inputs"]])
        except Exception:           # (a dry run cannot follow data-dependent control flow to the end)
            pass
        for mode in ex.scan_modes.values():
            total += 1
            if mode == "persistent":
                persistent += 1
            elif mode == "all-rows":            # no recurrence: one evaluation over whole sequences
                all_rows += 1
            else:
                want = expected_launch_list.get(c["name"])
                assert want is not None and want in mode, (c["name"], mode)
    assert (persistent, all_rows, total) == (95, 6, 105), (persistent, all_rows, total)
