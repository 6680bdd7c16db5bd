"""NumPy-2 compatibility monkey patches for the reference (pinned to numpy<2).

TEST INFRASTRUCTURE ONLY — must be imported before `aesara` in every oracle
process (see oracle/ref_overlay.py).  Restores the Python-level names the
reference touches at import/run time: tensor/type.py:104, tensor/elemwise.py:708
and :1427, tensor/basic.py:223, scalar/basic.py:3136.
"""
import numpy as np
import numpy.exceptions as _npe


def _obj2sctype(rep, default=None):
    try:
        return np.dtype(rep).type
    except Exception:
        return default


class _Cast:
    def __getitem__(self, t):
        return lambda x: np.asarray(x).astype(t)


_KIND_ORDER = ["b", "u", "i", "f", "c", "S", "U", "V", "O", "M", "m"]
_TEST_TYPES = "?bhilqpBHILQPefdgFDGO"


def _can_coerce_all(dtypes, start=0):
    if not dtypes:
        return None
    if len(dtypes) == 1:
        return dtypes[0]
    for ch in _TEST_TYPES[start:]:
        new = np.dtype(ch)
        if all(np.can_cast(x, new) for x in dtypes):
            return new
    return None


def _find_common_type(array_types, scalar_types):
    """NumPy 1.x ``np.find_common_type`` (removed in 2.0; used by the reference's
    tests/tensor/utils.py:150 only), restated from its documented algorithm: the smallest type
    every array type coerces to decides; the scalar types raise it only when their KIND is higher,
    and then to the smallest type at or above the scalar type that both coerce to."""
    arr = [np.dtype(t) for t in array_types]
    sca = [np.dtype(t) for t in scalar_types]
    maxa, maxsc = _can_coerce_all(arr), _can_coerce_all(sca)
    if maxa is None:
        return maxsc
    if maxsc is None:
        return maxa
    try:
        higher = _KIND_ORDER.index(maxsc.kind) > _KIND_ORDER.index(maxa.kind)
    except ValueError:
        return None
    if not higher:
        return maxa
    if np.can_cast(maxa, maxsc):
        return maxsc
    return _can_coerce_all([maxsc, maxa], start=_TEST_TYPES.index(maxsc.char))


for _name, _val in (
    ("find_common_type", _find_common_type),
    ("obj2sctype", _obj2sctype),
    ("sctype2char", lambda t: np.dtype(t).char),
    ("AxisError", _npe.AxisError),
    ("cast", _Cast()),
    ("ComplexWarning", _npe.ComplexWarning),
    ("VisibleDeprecationWarning", _npe.VisibleDeprecationWarning),
    ("float_", np.float64),
    ("complex_", np.complex128),
    ("product", np.prod),
    ("bool8", np.bool_),
    ("unicode_", np.str_),
    ("Inf", np.inf),
    ("NaN", np.nan),
    ("MAXDIMS", 32),   # NumPy 1.x value; only read as the "axis=None" sentinel of Op params
):
    if _name not in np.__dict__:
        try:
            setattr(np, _name, _val)
        except Exception:
            pass
