"""Horizontal fusion: independent full reductions of ONE scalar program — the
towers of a model, per-parameter norms — leave as one launch (``ahip_elemwise_reduce_all_multi``,
codegen ``spec.hjobs``): eager = replay = oracle, members of different shapes, more members than
one launch takes, members whose layout does not qualify, rebinding to new tensors on replay."""
import numpy as np
import pytest

import interp
from golden_util import CASES, case_expected, case_inputs, case_plan, assert_matches

pytestmark = pytest.mark.gpu


def _towers(shapes, dtype="float64", op="exp"):
    from aesara_amd.plan import Node, Plan
    p = Plan("towers", {}, [], [], [])
    for k, sh in enumerate(shapes):
        x = p.new_var(dtype, [None] * len(sh), "x%d" % k)
        e = p.new_var(dtype, [None] * len(sh))
        o = p.new_var("float64" if op == "exp" else dtype, [])
        p.inputs.append(x)
        p.outputs.append(o)
        if op == "exp":
            nodes = [{"op": "sqr", "in": [["i", 0]], "dtype": dtype},
                     {"op": "mul", "in": [["c", -0.5, dtype], ["t", 0]], "dtype": dtype},
                     {"op": "exp", "in": [["t", 1]], "dtype": dtype}]
        else:
            nodes = [{"op": "abs", "in": [["i", 0]], "dtype": dtype}]
        p.nodes.append(Node("Elemwise", [x], [e], {"scalar": {"n_in": 1, "nodes": nodes,
                                                              "out": [["t", len(nodes) - 1]]}}))
        p.nodes.append(Node("CAReduce", [e], [o], {
            "scalar_op": "add" if op == "exp" else "maximum", "axis": None,
            "acc_dtype": "float64" if op == "exp" else dtype}))
    return p


@pytest.mark.parametrize("name", ["hfuse_towers_f64", "hfuse_norms_f32"])
def test_golden_groups_run_as_one_launch(name):
    from aesara_amd.executor import PlanExecutor
    c = next(c for c in CASES if c["name"] == name)
    want = case_expected(c)
    for use_graph in (False, True):
        ex = PlanExecutor(case_plan(c), use_graph=use_graph)
        assert ex._hgroups, "no horizontal group found"
        for call in range(3):
            got = [o.cpu().numpy() for o in ex(*case_inputs(c))]
            assert_matches(c, got, want, "use_graph=%s call %d" % (use_graph, call))
        assert "ahip_elemwise_reduce_all_multi" in ex.trace
        ex.check()


def test_many_towers_of_different_shapes_against_the_oracle():
    """19 members (two launches: 16 + 3), shapes from 1 element to 1.5 M, one EMPTY member, one
    member whose input is a transposed view (runs alone: its layout is not flat)."""
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(3)
    shapes = [(1, 1), (7, 3), (64, 64), (1000, 1500), (0, 5), (33, 1), (4, 4096), (513, 129)] + \
        [(int(rng.integers(1, 300)), int(rng.integers(1, 300))) for _ in range(11)]
    plan = _towers(shapes)
    xs = [rng.standard_normal(sh) for sh in shapes]
    xs[6] = np.ascontiguousarray(xs[6].T).T                  # Fortran-ordered: a transposed view
    want = interp.run_plan(plan, xs)
    for use_graph in (False, True):
        ex = PlanExecutor(plan, use_graph=use_graph)
        dev = [torch.from_numpy(x).cuda() if x.flags.c_contiguous else
               torch.from_numpy(np.ascontiguousarray(x.T)).cuda().t() for x in xs]
        for call in range(2):
            got = ex(*dev)
            for k, (g, w) in enumerate(zip(got, want)):
                np.testing.assert_allclose(g.item(), w, rtol=1e-12, atol=1e-300, err_msg="tower %d" % k)
        assert ex.trace.count("ahip_elemwise_reduce_all_multi") >= 2
        # replay: new tensors of the same shapes are rebound, results follow the new data
        dev2 = [d * 0.5 for d in dev]
        want2 = interp.run_plan(plan, [x * 0.5 for x in xs])
        for g, w in zip(ex(*dev2), want2):
            np.testing.assert_allclose(g.item(), w, rtol=1e-12, atol=1e-300)
        ex.check()


def test_integer_and_max_groups_are_exact():
    import torch
    from aesara_amd.executor import PlanExecutor
    rng = np.random.default_rng(4)
    shapes = [(50, 80), (3, 3), (1024, 256), (9,), (16, 24)]     # (3, 3) and (9,): no 16-byte vectors
    plan = _towers(shapes, dtype="int32", op="max")
    xs = [rng.integers(-2 ** 30, 2 ** 30, sh).astype("int32") for sh in shapes]
    want = interp.run_plan(plan, xs)
    ex = PlanExecutor(plan)
    for call in range(2):
        got = ex(*[torch.from_numpy(x).cuda() for x in xs])
        for g, w in zip(got, want):
            assert g.item() == w
    assert "ahip_elemwise_reduce_all_multi" in ex.trace


def test_placed_towers_at_bench_shape_are_one_launch():
    """The bench's ``placed outputs`` row on one GPU: 8 independent fp64 2048^2 exp-sum towers in
    one plan = 256 MiB read by ONE launch (bench line r04: 43.6 us = 0.77 of the HBM peak; eight
    launches: 79 us = 0.42), results against an fp64 restatement."""
    import torch
    from aesara_amd.executor import PlanExecutor
    plan = _towers([(2048, 2048)] * 8)
    xs = [torch.randn(2048, 2048, dtype=torch.float64, device="cuda") for _ in range(8)]
    ex = PlanExecutor(plan, use_graph=True, borrow=True)
    for call in range(3):
        outs = ex(*xs)
        for x, o in zip(xs, outs):
            ref = torch.exp(-0.5 * x ** 2).sum().item()
            assert abs(o.item() - ref) <= 1e-11 * abs(ref)
    # (replay mode issues the launch twice on the first call: eager pass + recording pass)
    assert set(ex.trace) == {"ahip_elemwise_reduce_all_multi"} and len(ex.trace) == 2
