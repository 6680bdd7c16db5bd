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

from .devcell import DTYPE_NAMES as _DTYPE_NAMES
from .devcell import TORCH_DTYPES as _TORCH_DTYPES
from .devcell import DeviceCellMixin, DeviceFilterType, check_device_value  # noqa: F401  (re-exported)


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("aesara_amd: no HIP device is visible (device-resident values live "
                           "in HBM; pass device='cpu' only in host-logic tests)")
    return torch.device("cuda", torch.cuda.current_device())


class DeviceContainer(DeviceCellMixin, Container):
    """A storage cell (reference link/basic.py:39) that accepts device tensors as they are
    (logic: ``devcell.DeviceCellMixin``; host values take the reference's own filter path)."""

    def _host_set(self, value):
        Container.__set__(self, value)

    def __set__(self, value):
        self._device_set(value)

    data = property(Container.__get__, __set__)
    value = property(Container.__get__, __set__)

    @classmethod
    def adopt(cls, c: Container) -> "DeviceContainer":
        """Same type / storage list / flags as ``c`` (shares the storage cell)."""
        if isinstance(c, cls):
            return c
        return cls(c.type, c.storage, readonly=c.readonly, strict=c.strict,
                   allow_downcast=c.allow_downcast, name=c.name)

    @classmethod
    def adopt_input(cls, c: Container) -> "DeviceContainer":
        """``adopt`` for a cell ``Function.__call__`` filters arguments through
        (compile/function/types.py:853-863 ``s.type.filter(arg, ...)``): its ``type`` becomes the
        ``DeviceFilterType`` wrapper, which keeps device tensors on the device (no
        ``trust_input`` needed) and is the wrapped ``TensorType`` for everything else."""
        new = cls.adopt(c)
        if not isinstance(new.type, DeviceFilterType):
            new.type = DeviceFilterType(new.type)
        return new


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
