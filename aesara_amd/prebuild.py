"""Warm the in-tree generated-kernel cache (aesara_amd/_kcache) without a GPU.

hiprtc cross-compiles for gfx950 on a machine with no device, so the fused kernels that the
golden plans need can be generated at build time; the GPU box then only loads code objects.
Kernel *selection* depends on run-time shapes/strides, so this replays the executor's host
logic with a recording stub in place of launches (no compute happens here).
"""
from __future__ import annotations

import json
import os


def prebuild_golden_kernels(limit=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = os.path.join(root, "tests", "golden", "cases.json")
    if not os.path.exists(cases):
        return 0
    import sys

    from .executor import PlanExecutor
    from .plan import Plan

    tests = os.path.join(root, "tests")
    if tests not in sys.path:
        sys.path.insert(0, tests)
    from golden_inputs import make_input

    with open(cases) as f:
        data = json.load(f)["cases"]
    n = 0
    for c in data[:limit]:
        ex = PlanExecutor(Plan.from_json(c["plan"]), dry_run=True)
        try:
            ex(*[make_input(s) for s in c["inputs"]])
            n += 1
        except Exception as e:  # a dry run cannot follow data-dependent control flow
            print(f"prebuild: {c['name']}: {type(e).__name__}: {e}")
    n += _prebuild_full_shapes({c["name"]: c for c in data})
    return n


def _prebuild_full_shapes(cases):
    """The bench / smoke workloads at their real shapes select other kernel variants (flat
    16-byte-vector streams, longer rows) than the reduced golden shapes: generate those too so
    the first call on the GPU box only loads code objects."""
    import numpy as np

    from .device import DevArray, contiguous_strides
    from .executor import PlanExecutor, _FakeBuf
    from .plan import Plan

    def fake(shape, dtype):
        n = 1
        for s in shape:
            n *= s
        return DevArray(_FakeBuf(max(n, 1), dtype), 0, tuple(shape), contiguous_strides(shape), dtype)

    work = [
        ("cfg2_gauss_sum", [fake((4096, 4096), "float64"), np.float64(0.1), np.float64(1.3)]),
        ("cfg1b_matrix_add", [fake((4096, 4096), "float64"), fake((4096, 4096), "float64")]),
        ("cfg3b_gemm_update", [fake((512, 512), "float32")] * 3),
        ("cfg5_logistic", [fake((1 << 16, 256), "float32"), fake((256,), "float32"),
                           np.float32(0.1), fake((1 << 16,), "float32")]),
        # (the bench shapes proper: operands of 96 MiB and more take the streaming variants,
        # exec_elemwise.BIG_STREAM — 16-byte non-temporal loads)
        ("cfg5_logistic", [fake((1 << 24, 256), "float32"), fake((256,), "float32"),
                           np.float32(0.1), fake((1 << 24,), "float32")]),
        ("cfg3a_gemv", [fake((4096, 4096), "float64"), fake((4096,), "float64"), np.float64(2.0)]),
        ("softmax_rows_f32", [fake((4096, 1024), "float32")]),
    ]
    # BASELINE config 4 and its training step at the bench shape: the persistent Scan kernels are
    # specialised on the state length / batch (T is a run-time argument)
    T, H = 512, 1024
    Ws = [fake((H, H), "float32") for _ in range(6)]
    for name, B in (("cfg4_gru_b1_f32", 1), ("cfg4_gru_b8_f32", 64), ("gru_bptt_b1_f32", 1),
                    ("gru_bptt_b4_f32", 64)):
        x = fake((T, H) if B == 1 else (T, B, H), "float32")
        h0 = fake((H,) if B == 1 else (B, H), "float32")
        work.append((name, [x, h0] + Ws))
    done = 0
    for name, args in work:
        c = cases.get(name)
        if c is None:
            continue
        try:
            PlanExecutor(Plan.from_json(c["plan"]), dry_run=True)(*args)
            done += 1
        except Exception as e:
            print(f"prebuild: {name} (full shape): {type(e).__name__}: {e}")
    return done


if __name__ == "__main__":
    print("prebuilt kernels for", prebuild_golden_kernels(), "golden cases")
