"""CPU tier of ``tests/reference_files.py``: three of the reference's own test files, unmodified,
with the default mode = ``HipLinker`` over the ORACLE executor (checks the harness, the lowering of
every graph those files build and the outcome classification); the whole file list runs on the
MI355X in ``tests/test_gpu_reference_files.py``."""
import pytest

import reference_files as rf

import ref_overlay

if not ref_overlay.available():
    pytest.skip("no reference front end", allow_module_level=True)


def test_classification_rules():
    rep = {
        "tests/tensor/test_x.py::a": ["passed", ""],
        "tests/tensor/test_x.py::b": ["failed", "aesara_amd.lower.UnsupportedOp: dtype complex64 of x has no HIP kernels"],
        "tests/tensor/test_x.py::c": ["failed", "aesara_amd.lower.UnsupportedOp: Choose has no HIP lowering (outside ...)"],
        "tests/tensor/test_blas.py::TestGer::test_f32_0_0": ["failed", "AssertionError: Ger{destructive}"],
        "tests/tensor/test_x.py::d": ["failed", "AssertionError: values differ"],
        "tests/tensor/test_x.py::e": ["skipped", "slow"],
    }
    s, bad = rf.classify(rep)
    c = s["counts"]
    # "Choose has no HIP lowering" is NOT an explanation any more: only the Ops of the explicit
    # allow-list (reference_files.OUT_OF_SCOPE_OPS) are — a lowering that goes missing is unexplained
    assert (c["passed"], c["skipped"], c["out_of_scope"], c["not_applicable"], c["unexplained"]) == (1, 1, 1, 1, 2)
    assert sorted(bad) == ["tests/tensor/test_x.py::c", "tests/tensor/test_x.py::d"]
    rep2 = {"tests/tensor/test_x.py::p": ["passed", "", [2, 3]], "tests/tensor/test_x.py::q": ["passed", "", [0, 0]],
            "tests/tensor/test_x.py::r": ["failed", "aesara_amd.lower.UnsupportedOp: Pool has no HIP lowering (outside ...)", [0, 0]],
            "tests/tensor/test_x.py::k": ["failed", "AssertionError: Key not found in unpickled KeyData file.", [1, 0]]}
    s2, bad2 = rf.classify(rep2)
    c2 = s2["counts"]
    assert (c2["passed"], c2["through_hip"], c2["executed_hip"], c2["out_of_scope"], c2["environment"]) == (2, 1, 1, 1, 1)
    assert not bad2 and "through HipLinker 1" in rf.format_summary("oracle", s2, bad2).replace("THROUGH", "through")


def test_subtensor_special_and_shape_files_under_the_hip_mode():
    files = ["tests/tensor/test_subtensor.py", "tests/tensor/test_special.py", "tests/tensor/test_shape.py"]
    s, bad, text = rf.check("oracle", files, workers=4, timeout=1500)
    assert not bad, text
    c = s["counts"]
    assert c["passed"] >= 200, text                   # 142 + 42 + 33 collected; nearly all pass
    assert c["environment"] + c["out_of_scope"] + c["not_applicable"] <= 6, text
