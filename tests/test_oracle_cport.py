"""The oracle's C port (bench.py's cpu_baseline) must reproduce the reference's golden output
for BASELINE config 2 bit-for-bit (it restates the reference C linker's loops)."""
import numpy as np

import cport
from golden_util import CASES, case_expected, case_inputs


def test_cport_cfg2_matches_reference_golden():
    c = next(c for c in CASES if c["name"] == "cfg2_gauss_sum")
    x, mu, sg = case_inputs(c)
    assert cport.cfg2_eval(x, mu, sg) == float(case_expected(c)[0])


def test_cport_cfg1b():
    lib = cport.load()
    x, y = np.random.default_rng(0).random((2, 1000))
    out = np.empty(1000)
    lib.cport_cfg1b_add(x.ctypes.data, y.ctypes.data, out.ctypes.data, 1000)
    assert np.array_equal(out, x + y)


def test_cport_openmp_form_equals_serial():
    x = np.random.default_rng(5).standard_normal((257, 129))
    assert cport.cfg2_eval_omp(x, 0.1, 1.3, 4) == cport.cfg2_eval(x, 0.1, 1.3)
