"""Host cost of one `aesara.function(...)` call through the HIP linker (authoring container: the
reference front end is needed, no GPU): `Function.__call__` + this linker's thunk with the device
executor replaced by a stub that returns a preallocated result (what a replay hit costs on top
is measured on the GPU by tools/host_overhead.py).  Prints µs/call next to the reference's own
C-linker function on the same graph (BASELINE config 2 at a tiny shape)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402

import ref_overlay  # noqa: E402

ae = ref_overlay.import_reference()
import aesara.tensor as at  # noqa: E402
from aesara.compile.mode import Mode  # noqa: E402

from aesara_amd.linker import HIP_QUERY, HipLinker  # noqa: E402


def bench(f, args, n=20000):
    for _ in range(200):
        f(*args)
    t = time.perf_counter()
    for _ in range(n):
        f(*args)
    return (time.perf_counter() - t) / n * 1e6


def main():
    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    out = at.exp(-((x - mu) ** 2) / (2 * sg ** 2)).sum()
    xv = np.random.default_rng(1).standard_normal((64, 48))
    args = (xv, np.asarray(0.1), np.asarray(1.3))   # trust_input: exact ndarrays
    res = {}
    f_ref = ae.function([x, mu, sg], out, mode=Mode("cvm", "fast_run"))
    f_ref.trust_input = True
    res["reference C linker (64x48 compute included)"] = bench(f_ref, args)
    result = [np.float64(0.0)]
    for fast in (False, True):
        lk = HipLinker(executor_factory=lambda plan: (lambda *a: result), fast_call=fast)
        f = ae.function([x, mu, sg], out, mode=Mode(lk, HIP_QUERY))
        f.trust_input = True
        res["HipLinker thunk, stub executor, fast_call=%s" % fast] = bench(f, args)
    for k, v in res.items():
        print("%-60s %7.2f us/call" % (k, v))


if __name__ == "__main__":
    main()
