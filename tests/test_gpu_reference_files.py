"""The reference's own test FILES, unmodified, with the default mode = HIP on the MI355X
(``tests/reference_files.py``: test_subtensor / test_basic / test_blas / test_special /
test_extra_ops / test_elemwise / test_shape / test_math / scan/test_basic — every test of those
files through ``aesara.function`` -> ``HipLinker`` -> ``PlanExecutor`` -> the C-ABI; a child
``pytest`` per call, several workers sharing the device).  A test that does not pass must be
explained (fails with the reference's own linker in this image / UnsupportedOp for an out-of-scope
dtype or Op / listed not-applicable); anything else fails here.  The summary is written to
``gpurun_out/r06_reference_files.log`` when that directory can be created (copied to ``profiles/``).

What the run PROVES is the count of passing tests that compiled at least one function through
``HipLinker`` (``through_hip``: the plugin counts ``jit_compile`` calls and executor calls per test)
— a test that builds no function, or names its own linker, passes without touching the HIP path and
is not counted.  The assertion below is on that number."""
import os

import pytest

import reference_files as rf

import ref_overlay

pytestmark = pytest.mark.gpu

if not ref_overlay.available():
    pytest.skip("no reference front end (oracle/_ref overlay not packed)", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_test_files_under_the_hip_mode_on_the_device():
    import torch
    assert torch.cuda.is_available()
    workers = max(2, min(12, (os.cpu_count() or 4) // 4))
    log = os.path.join(ROOT, "gpurun_out", "r06_reference_files.log")
    s, bad, text = rf.check("device", rf.FILES, workers=workers, log_path=log, timeout=3300)
    print(text)
    assert not bad, text
    c = s["counts"]
    # round 5's file list: 967 tests through HipLinker; round 6 adds test_ifelse / test_sort /
    # rewriting/test_elemwise / test_checkpoints / test_raise_op / test_updates / test_xlogx /
    # test_math_scipy / compile/test_ops / test_inplace and test_keepdims with its Mode re-pointed
    # (+ the second batch: tensor/rewriting/test_{basic,math,subtensor,shape,special,uncanonicalize,extra_ops},
    # compile/test_builders, scalar/test_{basic,math}, nnet/test_sigm: 331 more through HipLinker)
    assert c["through_hip"] >= 2500, text
    assert c["executed_hip"] >= 2350, text
