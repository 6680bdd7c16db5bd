"""CPU run of the reference's own backend-parameterised test classes (tests/reference_suites.py)
through ``HipLinker`` over the ORACLE executor: checks the harness and that every graph those
suites build lowers to a plan (the numerics here are the oracle's; the HIP kernels run the same
classes in tests/test_gpu_reference_suites.py)."""
import pytest

import reference_suites as rs

if not rs.available():
    pytest.skip("no reference front end", allow_module_level=True)

globals().update(rs.build(real_device=False))
