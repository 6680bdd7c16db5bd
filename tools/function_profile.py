#!/usr/bin/env python
"""Where the host time of one ``f(x, mu, sigma)`` goes (config 2 through aesara.function(mode="HIP"),
untrusted Function.__call__): cProfile over N calls with the device kept busy.
usage (GPU box): python tools/function_profile.py [--trust] [--small] > gpurun_out/function_profile.txt"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    import ref_overlay
    ae = ref_overlay.import_reference()
    import aesara.tensor as at
    import aesara_amd
    aesara_amd.get_mode()
    x, mu, sg = at.dmatrix("x"), at.dscalar("mu"), at.dscalar("sigma")
    f = ae.function([x, mu, sg], at.exp(-(x - mu) ** 2 / (2 * sg ** 2)).sum(), mode="HIP")
    f.trust_input = "--trust" in sys.argv
    # --small: 256 x 256 matrices — the device is never the bottleneck, the loop measures the host alone
    side = 256 if "--small" in sys.argv else 4096
    xs = [torch.randn(side, side, dtype=torch.float64, device="cuda") for _ in range(8)]
    m, s = np.asarray(0.1), np.asarray(1.3)
    for k in range(64):
        f(xs[k % 8], m, s)
    torch.cuda.synchronize()
    N = 3000
    t0 = time.perf_counter()
    for k in range(N):
        f(xs[k % 8], m, s)
    t_host = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / N * 1e6
    print("host us/call (issue only) %.2f; us/call incl. device %.2f; trust_input=%s" % (t_host, t_all, f.trust_input))
    pr = cProfile.Profile()
    pr.enable()
    for k in range(N):
        f(xs[k % 8], m, s)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
