"""Device-resident storage cells — the part of ``sharedvar.DeviceContainer`` that does not need
the reference front end (so it also runs, and is tested, on a GPU box without Aesara).

``Function.__call__`` writes ``updates=`` results back with ``storage.data = value``
(reference compile/function/types.py:1060-1069), i.e. ``Container.__set__`` -> ``Type.filter``
(link/basic.py:93-119); ``TensorType.filter`` is ``np.asarray`` (tensor/type.py:135).  A
:class:`DeviceCellMixin` cell keeps a value that is already a typed, shaped device tensor as it is
(after the dtype / rank / static-shape checks of that filter) and only sends host data through
the host filter, uploading it once when the cell has a target device.
"""
from __future__ import annotations

import numpy as np
import torch

TORCH_DTYPES = {
    "float32": torch.float32, "float64": torch.float64, "int8": torch.int8,
    "int16": torch.int16, "int32": torch.int32, "int64": torch.int64,
    "uint8": torch.uint8, "bool": torch.bool,
}
DTYPE_NAMES = {v: k for k, v in TORCH_DTYPES.items()}


def check_device_value(typ, data):
    """The checks of ``TensorType.filter`` (tensor/type.py:135-256) that apply to a value that
    is already a typed, shaped device tensor: dtype, rank, static shape.  ``typ`` needs
    ``dtype`` (name), ``ndim`` and ``shape`` (tuple with ``None`` for unknown extents)."""
    want = TORCH_DTYPES.get(typ.dtype)
    if want is None or data.dtype != want:
        raise TypeError(f"{typ}: device tensor has dtype {data.dtype}, expected {typ.dtype}")
    if data.ndim != typ.ndim:
        raise TypeError(f"Wrong number of dimensions: expected {typ.ndim}, "
                        f"got {data.ndim} with shape {tuple(data.shape)}.")
    if not all(s is None or s == ds for s, ds in zip(typ.shape, data.shape)):
        raise TypeError(f"The type's shape ({typ.shape}) is not compatible with the "
                        f"data's ({tuple(data.shape)})")
    return data


class DeviceCellMixin:
    """``__set__`` of a storage cell that accepts device tensors as they are.  The host class
    supplies ``type``, ``storage`` (one-element list), ``readonly``, ``name`` and
    ``_host_set(value)`` (the reference's ``Container.__set__``)."""

    device = None   # None: host values stay ndarrays (plain inputs); else upload target

    def _device_set(self, value):
        if isinstance(value, torch.Tensor):
            if self.readonly:
                raise Exception(f"Cannot set readonly storage: {self.name}")
            try:
                self.storage[0] = check_device_value(self.type, value)
            except Exception as e:
                e.args = e.args + (f'Container name "{self.name}"',)
                raise
            return
        self._host_set(value)
        if self.device is not None and isinstance(self.storage[0], np.ndarray):
            # (np.ascontiguousarray would turn a 0-d value — a scalar parameter — into shape (1,))
            self.storage[0] = torch.from_numpy(
                np.array(self.storage[0], order="C", copy=True)).to(self.device)


class PlainType:
    """Minimal stand-in for ``TensorType`` (dtype name, rank, static shape, host filter)."""

    def __init__(self, dtype, shape):
        self.dtype, self.shape, self.ndim = dtype, tuple(shape), len(shape)

    def filter(self, value, strict=False, allow_downcast=None):
        a = np.asarray(value)
        if a.dtype.name != self.dtype:
            if strict:
                raise TypeError(f"expected {self.dtype}, got {a.dtype.name}")
            a = a.astype(self.dtype)
        if a.ndim != self.ndim:
            raise TypeError(f"Wrong number of dimensions: expected {self.ndim}, got {a.ndim}")
        return a

    def __repr__(self):
        return f"PlainType({self.dtype}, {self.shape})"


class DeviceCell(DeviceCellMixin):
    """Stand-alone cell with the interface ``Function.__call__`` uses (``.storage``, ``.data``,
    ``.value``): what ``sharedvar.DeviceContainer`` is without the reference base class."""

    def __init__(self, typ, storage=None, readonly=False, strict=False, allow_downcast=None,
                 name=None, device=None):
        self.type, self.storage = typ, storage if storage is not None else [None]
        self.readonly, self.strict, self.allow_downcast, self.name = readonly, strict, allow_downcast, name
        self.device = device

    def _host_set(self, value):
        if self.readonly:
            raise Exception(f"Cannot set readonly storage: {self.name}")
        self.storage[0] = None if value is None else self.type.filter(
            value, strict=self.strict, allow_downcast=self.allow_downcast)

    def __get__(self):
        return self.storage[0]

    data = property(__get__, DeviceCellMixin._device_set)
    value = property(__get__, DeviceCellMixin._device_set)
