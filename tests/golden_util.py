"""Helpers shared by the oracle and GPU parity tests: load golden cases, compare outputs."""
import json
import os

import numpy as np

from aesara_amd.plan import Plan
from golden_inputs import make_input

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)["cases"]


CASES = load_cases()
CASE_IDS = [c["name"] for c in CASES]


def case_plan(c):
    return Plan.from_json(c["plan"])


def case_inputs(c):
    return [make_input(s) for s in c["inputs"]]


def case_expected(c):
    z = np.load(os.path.join(GOLDEN, c["name"] + ".npz"))
    return [z[f"out{k}"] for k in range(c["n_out"])]


def assert_matches(c, got, expected, what="output"):
    assert len(got) == len(expected)
    for k, (g, e) in enumerate(zip(got, expected)):
        g = np.asarray(g)
        assert g.shape == e.shape, f"{c['name']} {what} {k}: shape {g.shape} != {e.shape}"
        assert g.dtype == e.dtype, f"{c['name']} {what} {k}: dtype {g.dtype} != {e.dtype}"
        if c["exact"]:
            # bit-exact bar for integer / index / comparison work
            assert np.array_equal(g, e, equal_nan=True), f"{c['name']} {what} {k}: not bit-exact"
        else:
            np.testing.assert_allclose(g, e, rtol=c["rtol"], atol=c["atol"], equal_nan=True,
                                       err_msg=f"{c['name']} {what} {k}")
