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


for _name, _val in (
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
