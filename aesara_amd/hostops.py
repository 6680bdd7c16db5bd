"""Host-side glue ops: integer shape / index arithmetic stays on the CPU.

The reference runs ``Shape_i``, ``ScalarFromTensor``, ``MakeVector`` and the 0-d int64
``Elemwise{Composite{Switch(LT…)}}`` nodes that Scan/Subtensor lowering emits as ordinary
thunks (tensor/shape.py:189, tensor/basic.py:594/539/1629).  They touch a handful of integers,
so the HIP linker evaluates them on the host with exact NumPy integer semantics instead of
launching kernels; nothing here ever sees array data that lives on the device.
"""
from __future__ import annotations

import numpy as np

_CMP = {"lt": np.less, "gt": np.greater, "le": np.less_equal, "ge": np.greater_equal,
        "eq": np.equal, "neq": np.not_equal}
_NARY = {"add": np.add, "mul": np.multiply, "maximum": np.maximum, "minimum": np.minimum,
         "and": np.bitwise_and, "or": np.bitwise_or, "xor": np.bitwise_xor}
_BIN = {"sub": np.subtract, "int_div": np.floor_divide, "mod": np.mod, "pow": np.power,
        "true_div": np.true_divide}
_UN = {"neg": np.negative, "abs": np.abs, "sgn": np.sign, "sqr": np.square,
       "identity": lambda x: x, "invert": np.invert, "ceil": np.ceil, "floor": np.floor,
       "trunc": np.trunc, "sqrt": np.sqrt, "exp": np.exp, "log": np.log,
       "reciprocal": np.reciprocal, "round_half_to_even": np.around}


_OTHER = ("cast", "second", "switch", "clip")


def host_evaluable(scalar):
    """Every op of the expression is one the host glue restates (anything else — arctan2 of two
    ScalarType inputs, special functions — runs as a kernel on 0-d device values)."""
    return all(n["op"] in _NARY or n["op"] in _CMP or n["op"] in _BIN or n["op"] in _UN or n["op"] in _OTHER
               for n in scalar["nodes"])


_HOST_MEMO = {}


def eval_scalar_host(scalar, ins):
    """Evaluate a plan scalar expression on small host arrays (shape arithmetic only).  The same
    shape arithmetic recurs on every call of a function, so results are memoised on the
    expression's identity and the (tiny) input values."""
    arrs = [np.asarray(x) for x in ins]
    sig = None
    if all(a.size <= 8 for a in arrs):
        sig = (id(scalar),) + tuple((a.dtype.str, a.shape, a.tobytes()) for a in arrs)
        hit = _HOST_MEMO.get(sig)
        if hit is not None:
            return hit[0]
    res = _eval_scalar_host(scalar, arrs)
    if sig is not None:
        if len(_HOST_MEMO) > 4096:
            _HOST_MEMO.clear()
        _HOST_MEMO[sig] = (res, scalar)     # keeps `scalar` alive: its id stays unique
    return res


def _eval_scalar_host(scalar, ins):
    temps = []

    def get(r):
        if r[0] == "i":
            return ins[r[1]]
        if r[0] == "t":
            return temps[r[1]]
        return np.asarray(r[1], dtype=r[2])

    with np.errstate(all="ignore"):
        for n in scalar["nodes"]:
            op, dt = n["op"], np.dtype(n["dtype"])
            a = [np.asarray(get(r)) for r in n["in"]]
            if dt.kind == "f" and (op in _UN or op in _BIN):
                # a float result of integer operands (upgrade_to_float ops, true_div): the reference's C
                # body computes in the OUTPUT type (scalar/basic.py c_code: operands convert on use) —
                # NumPy would pick float16 for int8 and an integer reciprocal
                a = [x.astype(dt) if x.dtype.kind in "iub" else x for x in a]
            if op in _NARY:
                r = a[0]
                for x in a[1:]:
                    r = _NARY[op](r, x)
            elif op in _CMP:
                r = _CMP[op](a[0], a[1])
            elif op in _BIN:
                r = _BIN[op](a[0], a[1])
            elif op in _UN:
                r = _UN[op](a[0])
            elif op == "cast":
                r = a[0]
            elif op == "second":
                r = np.broadcast_arrays(a[0], a[1])[1]
            elif op == "switch":
                r = np.where(a[0] != 0, a[1], a[2])
            elif op == "clip":
                r = np.where(a[0] < a[1], a[1], np.where(a[0] > a[2], a[2], a[0]))
            else:
                raise NotImplementedError(f"host glue: scalar op {op}")
            temps.append(np.asarray(r).astype(dt))
    return [np.asarray(get(r)) for r in scalar["out"]]


def resolve_index(idx_list, extra):
    """Plan ``idx_list`` + dynamic host ints -> tuple of python slices / ints
    (reference: tensor/subtensor.py:756 Subtensor.perform via get_idx_list)."""
    extra = list(extra)

    def ent(e):
        if e == "in":
            return int(extra.pop(0))
        return e

    out = []
    for e in idx_list:
        if "slice" in e:
            st, sp, se = (ent(t) for t in e["slice"])
            out.append(slice(st, sp, se))
        else:
            out.append(ent(e["index"]))
    if extra:
        raise ValueError("unused dynamic index inputs")
    return tuple(out)


def view_from_index(shape, strides, offset, index):
    """Apply basic (slice / int) indexing to a strided view described by
    (shape, strides[elements], offset[elements]) — pure stride arithmetic, no data access.
    Raises IndexError like NumPy for out-of-range integers."""
    shape, strides = list(shape), list(strides)
    nshape, nstrides = [], []
    if len(index) > len(shape):
        raise IndexError("too many indices for array")
    for d, ix in enumerate(index):
        n = shape[d]
        if isinstance(ix, slice):
            start, stop, step = ix.indices(n)
            length = len(range(start, stop, step))
            offset += start * strides[d] if length > 0 else 0
            nshape.append(length)
            nstrides.append(strides[d] * step)
        else:
            i = int(ix)
            if i < -n or i >= n:
                raise IndexError(f"index {i} is out of bounds for axis {d} with size {n}")
            if i < 0:
                i += n
            offset += i * strides[d]
    for d in range(len(index), len(shape)):
        nshape.append(shape[d])
        nstrides.append(strides[d])
    return tuple(nshape), tuple(nstrides), offset
