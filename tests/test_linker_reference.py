"""The drop-in boundary against the REAL reference front end (authoring container only; skipped
on the GPU box where /root/reference is absent): HipLinker plugs into aesara.function through
the JITLinker API, and the plans committed under tests/golden are exactly what it lowers."""
import json

import numpy as np
import pytest

import ref_overlay

pytestmark = pytest.mark.skipif(not ref_overlay.available(),
                                reason="reference Aesara not present (GPU box)")


@pytest.fixture(scope="module")
def ae():
    return ref_overlay.import_reference()


def _oracle_linker():
    import interp
    from aesara_amd.linker import HipLinker
    return HipLinker(executor_factory=lambda plan: (lambda *a: interp.run_plan(plan, a)))


def test_mode_registration(ae):
    from aesara.compile.mode import get_mode, predefined_linkers
    import aesara_amd
    mode = aesara_amd.get_mode()
    assert "hip" in predefined_linkers
    assert get_mode("HIP") is mode
    from aesara.link.basic import JITLinker
    assert isinstance(mode.linker, JITLinker)


def test_function_through_jitlinker_matches_c_linker(ae):
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    y = at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum()
    f_hip = ae.function([x, mu, sg], y, mode=Mode(_oracle_linker(), HIP_QUERY))
    f_ref = ae.function([x, mu, sg], y)  # reference default: cvm + C thunks
    xv = np.random.default_rng(1).standard_normal((64, 48))
    np.testing.assert_allclose(f_hip(xv, 0.1, 1.3), f_ref(xv, 0.1, 1.3), rtol=1e-12)
    # README example (README.md:49-60): a + b on dscalars
    a, b = at.dscalar("a"), at.dscalar("b")
    f = ae.function([a, b], a + b, mode=Mode(_oracle_linker(), HIP_QUERY))
    assert f(1.5, 2.5) == 4.0


def test_shared_variable_updates(ae):
    """check_blas.py:54-57 pattern: updates=[(C, 0.4*C + 0.8*dot(A,B))] — the linker returns the
    update expression as an extra output and Function stores it back."""
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    rng = np.random.default_rng(0)
    A, B = (ae.shared(rng.standard_normal((8, 8)).astype("float32")) for _ in range(2))
    Cs = ae.shared(np.zeros((8, 8), "float32"))
    f = ae.function([], [], updates=[(Cs, np.float32(0.4) * Cs + np.float32(0.8) * at.dot(A, B))],
                    mode=Mode(_oracle_linker(), HIP_QUERY))
    f(); f()
    ab = A.get_value() @ B.get_value()
    np.testing.assert_allclose(Cs.get_value(), 0.4 * 0.8 * ab + 0.8 * ab, rtol=1e-5)
    ops = [n.op for n in f.maker.linker.plan.nodes]
    assert ops == ["Gemm"]


def test_linker_clone_and_scan_inner_mode(ae):
    """Linker.clone(allow_gc=…) is used by Scan/Mode.clone (link/basic.py:190)."""
    from aesara_amd.linker import HipLinker
    l2 = HipLinker().clone(allow_gc=False)
    assert isinstance(l2, HipLinker) and l2.allow_gc is False


def test_unsupported_op_fails_loudly(ae):
    import aesara.tensor as at
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    from aesara_amd.lower import UnsupportedOp
    x = at.dmatrix("x")
    with pytest.raises(UnsupportedOp):
        ae.function([x], at.gammaln(x), mode=Mode(_oracle_linker(), HIP_QUERY))


@pytest.mark.parametrize("name", ["cfg2_gauss_sum", "cfg3b_gemm_update", "cfg5_logistic",
                                  "cfg4_gru_b1_f32", "red7_f64", "subtensor_basic"])
def test_committed_plans_are_what_the_linker_lowers(ae, name):
    """Re-lower the graph with the live reference and compare with the committed plan JSON."""
    import gen_golden
    from aesara.compile.mode import Mode
    from aesara_amd.linker import HIP_QUERY
    from golden_util import CASES
    fn = next(f for n, f, *_ in gen_golden.CASES if n == name)
    ins, outs, _ = fn()
    f = ae.function(ins, outs, mode=Mode(_oracle_linker(), HIP_QUERY), on_unused_input="ignore")
    plan = f.maker.linker.plan
    plan.name = name
    committed = next(c for c in CASES if c["name"] == name)["plan"]
    assert json.dumps(plan.to_json(), sort_keys=True) == json.dumps(committed, sort_keys=True)
