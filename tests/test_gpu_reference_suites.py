"""The reference's own backend-parameterised test classes under ``mode="HIP"`` on the MI355X
(tests/reference_suites.py: TestBroadcast, TestCAReduce, TestDimShuffle, TestGemm.cmp, TestSgemv /
TestDgemv, TestBlasStrides, the restatement tests of TestScan / TestExamples) — the real
``PlanExecutor``, results converted to ndarrays for the reference's own assertions."""
import pytest

import reference_suites as rs

pytestmark = pytest.mark.gpu

if not rs.available():
    pytest.skip("no reference front end (oracle/_ref overlay not packed)", allow_module_level=True)

import torch  # noqa: E402

if torch.cuda.is_available():
    globals().update(rs.build(real_device=True))
else:
    @pytest.mark.gpu
    def test_needs_a_device():
        pytest.skip("no HIP device")
