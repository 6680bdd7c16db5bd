"""Deterministic synthetic inputs shared by the golden generator (oracle/gen_golden.py, which
runs the reference in the authoring container) and the parity tests (which run on the GPU box
where the reference is absent).  Only NumPy; same NumPy version in both places (same image).
"""
import numpy as np


def make_input(spec):
    """spec: {"kind", "seed", "shape", "dtype", ...} -> ndarray (possibly a non-contiguous view
    when spec["view"] is given, to exercise stride handling)."""
    rng = np.random.default_rng(spec.get("seed", 0))
    shape = tuple(spec["shape"])
    dt = np.dtype(spec["dtype"])
    base_shape = shape
    view = spec.get("view")
    if view:
        if view["kind"] == "transpose":
            base_shape = tuple(shape[::-1])
        elif view["kind"] == "step":
            base_shape = tuple(s * view["step"] for s in shape)
    kind = spec["kind"]
    if kind == "normal":
        a = rng.standard_normal(base_shape) * spec.get("scale", 1.0) + spec.get("shift", 0.0)
    elif kind == "uniform":
        a = rng.uniform(spec.get("low", 0.0), spec.get("high", 1.0), base_shape)
    elif kind == "randint":
        a = rng.integers(spec["low"], spec["high"], base_shape)
    elif kind == "bernoulli":
        a = rng.random(base_shape) < spec.get("p", 0.5)
    elif kind == "normal_with_nan":
        # a few NaNs and repeated maxima: argmax / max NaN and tie handling
        a = np.round(rng.standard_normal(base_shape) * 2.0)
        flat = a.reshape(-1)
        flat[rng.integers(0, flat.size, 3) + flat.size // 2 - flat.size // 2] = np.nan
    elif kind == "const_list":
        a = np.asarray(spec["values"]).reshape(base_shape)
    elif kind == "const":
        a = np.full(base_shape, spec["value"])
    elif kind == "perm":
        a = rng.permutation(spec["n"])[: int(np.prod(base_shape))].reshape(base_shape)
    elif kind == "arange":
        a = np.arange(int(np.prod(base_shape))).reshape(base_shape) * spec.get("scale", 1)
    else:
        raise ValueError(kind)
    a = np.array(np.asarray(a).astype(dt), order="C", copy=True)
    if view:
        if view["kind"] == "transpose":
            a = a.T
        elif view["kind"] == "step":
            a = a[tuple(slice(None, None, view["step"]) for _ in shape)]
    assert a.shape == shape, (a.shape, shape)
    return a
