"""Device-resident state for the HIP linker: ``DeviceContainer`` and ``hip_shared``.

SURVEY §8(f).2.  The reference keeps the state of a training loop in ``SharedVariable``s whose
``updates=`` are written back by ``Function.__call__`` with ``storage.data = output``
(reference compile/function/types.py:1060-1069), i.e. ``Container.__set__`` -> ``Type.filter``
(link/basic.py:93-119).  ``TensorType.filter`` is ``np.asarray`` (tensor/type.py:135), so the
``check_blas.py:54-57`` pattern ``updates=[(C, 0.4*C + 0.8*dot(A, B))]`` would move C over PCIe
twice per call.  A new ``Type`` is not an option — the reference's dense-op checks are
``type(t) == TensorType`` (``DenseTypeMeta.__instancecheck__`` tensor/type.py:632) — so the
*type* stays the plain ``TensorType`` and the *cell* changes: a ``DeviceContainer`` validates a
device tensor (dtype, rank, static shape) and keeps it as is; host data still goes through the
reference's own filter and is uploaded once.  ``HipLinker.make_all`` hands such containers to
``Function``, so update outputs are stored back without leaving HBM.

Needs the reference front end (``import aesara``) and torch as the device container; no compute.
"""
from __future__ import annotations

import numpy as np
import torch

from aesara.link.basic import Container
from aesara.tensor.sharedvar import TensorSharedVariable
from aesara.tensor.type import TensorType

_TORCH_DTYPES = {
    "float32": torch.float32, "float64": torch.float64, "int8": torch.int8,
    "int16": torch.int16, "int32": torch.int32, "int64": torch.int64,
    "uint8": torch.uint8, "bool": torch.bool,
}
_DTYPE_NAMES = {v: k for k, v in _TORCH_DTYPES.items()}


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("aesara_amd: no HIP device is visible (device-resident values live "
                           "in HBM; pass device='cpu' only in host-logic tests)")
    return torch.device("cuda", torch.cuda.current_device())


def check_device_value(typ, data):
    """The checks of ``TensorType.filter`` (tensor/type.py:135-256) that apply to a value that
    is already a typed, shaped device tensor: dtype, rank, static shape."""
    want = _TORCH_DTYPES.get(typ.dtype)
    if want is None or data.dtype != want:
        raise TypeError(f"{typ}: device tensor has dtype {data.dtype}, expected {typ.dtype}")
    if data.ndim != typ.ndim:
        raise TypeError(f"Wrong number of dimensions: expected {typ.ndim}, "
                        f"got {data.ndim} with shape {tuple(data.shape)}.")
    if not all(s is None or s == ds for s, ds in zip(typ.shape, data.shape)):
        raise TypeError(f"The type's shape ({typ.shape}) is not compatible with the "
                        f"data's ({tuple(data.shape)})")
    return data


class DeviceContainer(Container):
    """A storage cell (reference link/basic.py:39) that accepts device tensors as they are."""

    device = None   # None: host values stay ndarrays (plain inputs); else upload target

    def __set__(self, value):
        if isinstance(value, torch.Tensor):
            if self.readonly:
                raise Exception(f"Cannot set readonly storage: {self.name}")
            try:
                self.storage[0] = check_device_value(self.type, value)
            except Exception as e:
                e.args = e.args + (f'Container name "{self.name}"',)
                raise
            return
        Container.__set__(self, value)
        if self.device is not None and isinstance(self.storage[0], np.ndarray):
            self.storage[0] = torch.from_numpy(
                np.ascontiguousarray(self.storage[0])).to(self.device)

    data = property(Container.__get__, __set__)
    value = property(Container.__get__, __set__)

    @classmethod
    def adopt(cls, c: Container) -> "DeviceContainer":
        """Same type / storage list / flags as ``c`` (shares the storage cell)."""
        if isinstance(c, cls):
            return c
        return cls(c.type, c.storage, readonly=c.readonly, strict=c.strict,
                   allow_downcast=c.allow_downcast, name=c.name)


class HipTensorSharedVariable(TensorSharedVariable):
    """Shared variable whose cell holds a device tensor (reference tensor/sharedvar.py:22,
    compile/sharedvalue.py:30).  Its type is the ordinary ``TensorType``."""

    def get_value(self, borrow=False, return_internal_type=False):
        """Host ``ndarray`` by default (what the reference returns); the device tensor itself
        with ``return_internal_type=True`` (``borrow=True``: no copy)."""
        v = self.container.value
        if return_internal_type:
            return v if borrow else v.clone()
        return v.detach().cpu().numpy()

    def set_value(self, new_value, borrow=False):
        if isinstance(new_value, torch.Tensor) and not borrow:
            new_value = new_value.clone()
        self.container.value = new_value     # DeviceContainer: check or filter + upload

    def zero(self, borrow=False):
        self.container.value.zero_()


def hip_shared(value, name=None, strict=False, allow_downcast=None, shape=None, device=None):
    """Counterpart of ``aesara.shared`` (reference tensor/sharedvar.py:48 ``tensor_constructor``)
    for state that stays in HBM across calls::

        C = hip_shared(np.zeros((4096, 4096), "float32"))
        f = aesara.function([], [], updates=[(C, 0.4 * C + 0.8 * dot(A, B))], mode="HIP")
    """
    dev = torch.device(device) if device is not None else _default_device()
    if isinstance(value, torch.Tensor):
        dtype, vshape = _DTYPE_NAMES[value.dtype], tuple(value.shape)
        value = value.to(dev)
    else:
        value = np.asarray(value)
        dtype, vshape = value.dtype.name, value.shape
    if dtype not in _TORCH_DTYPES:
        raise TypeError(f"hip_shared: dtype {dtype} has no device representation")
    typ = TensorType(dtype, shape=(None,) * len(vshape) if shape is None else shape)
    cell = DeviceContainer(typ, [None], readonly=False, strict=strict,
                           allow_downcast=allow_downcast, name=name)
    cell.device = dev
    cell.value = value
    return HipTensorSharedVariable(type=typ, value=None, strict=None, container=cell, name=name)


__all__ = ["DeviceContainer", "HipTensorSharedVariable", "hip_shared"]
