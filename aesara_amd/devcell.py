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


_CASTERS = {}


def _device_cast(data, dtype):
    """``data`` (device tensor) converted to ``dtype`` by the generated cast kernel (the device
    counterpart of the ``_asarray(data, dtype)`` in ``TensorType.filter``): a one-node Elemwise
    plan through the HIP executor, cached per (source dtype, target dtype, rank)."""
    from .executor import PlanExecutor
    from .plan import Plan

    src = DTYPE_NAMES[data.dtype]
    key = (src, dtype, data.ndim)
    ex = _CASTERS.get(key)
    if ex is None:
        shape = [None] * data.ndim
        ex = _CASTERS[key] = PlanExecutor(Plan.from_json({
            "version": 1, "name": "filter_cast_%s_%s" % (src, dtype),
            "vars": [{"id": 0, "dtype": src, "shape": shape}, {"id": 1, "dtype": dtype, "shape": shape}],
            "inputs": [0], "outputs": [1],
            "nodes": [{"op": "Elemwise", "inputs": [0], "outputs": [1], "params": {"scalar": {
                "n_in": 1, "nodes": [{"op": "cast", "in": [["i", 0]], "dtype": dtype}],
                "out": [["t", 0]]}}}]}), use_graph=False)
    (out,) = ex(data)
    return out


def filter_device_value(typ, data, strict=False, allow_downcast=None):
    """``TensorType.filter`` (tensor/type.py:135-256) for a value that already is a device tensor:
    the same decisions, taken without bringing the data to the host.  ``strict``: the dtype must
    be the type's (:163-173).  Otherwise a value whose dtype the type can represent exactly is
    converted (on the device), any other dtype only with ``allow_downcast`` (:175-203, same
    message); then the rank / static-shape checks (:236-251)."""
    src = DTYPE_NAMES.get(data.dtype)
    if src is None:
        raise TypeError(f"{typ}: device tensors of dtype {data.dtype} have no counterpart here")
    if src != typ.dtype:
        if strict:
            raise TypeError(f"{typ} expected a tensor with dtype={typ.dtype} (got {src}).")
        if not allow_downcast and np.promote_types(src, typ.dtype).name != typ.dtype:
            raise TypeError(
                f"{typ} cannot store a value of dtype {src} without risking loss of precision. "
                f"If you do not mind this loss, you can: 1) explicitly cast your data to "
                f'{typ.dtype}, or 2) set "allow_input_downcast=True" when calling "function". '
                f"Value: device tensor of shape {tuple(data.shape)}")
        if data.ndim != typ.ndim:
            raise TypeError(f"Wrong number of dimensions: expected {typ.ndim}, "
                            f"got {data.ndim} with shape {tuple(data.shape)}.")
        data = _device_cast(data, typ.dtype)
    return check_device_value(typ, data)


class DeviceFilterType:
    """What the input cells of a HIP-linked ``Function`` carry as ``.type``.

    ``Function.__call__`` filters every positional argument with
    ``s.type.filter(arg, strict=s.strict, allow_downcast=s.allow_downcast)`` where ``s`` is the
    input *container* (compile/function/types.py:853-863) — and ``TensorType.filter`` starts
    with ``np.asarray``, which a device tensor cannot survive.  The graph-level ``Type`` has to
    stay the plain ``TensorType`` (``DenseTypeMeta``, tensor/type.py:632), so the container's type
    is wrapped instead: ``filter`` keeps a device tensor on the device (``filter_device_value``:
    same strict / downcast / rank / shape decisions) and sends everything else to the wrapped
    type; every other attribute, equality and hashing are the wrapped type's.

    The aliased-input copy of ``Function.__call__`` (:898-938) asks the *variable's* type —
    ``TensorType.may_share_memory`` is False for anything but two ndarrays — so two device
    arguments over the same memory are passed as they are.  That check protects inputs an
    in-place Op would destroy; this linker's rewrite query excludes ``inplace`` and the executor
    only ever writes into buffers it allocated itself, so there is nothing for it to protect.
    """
    __slots__ = ("_t", "_want", "_ndim", "_static", "_np")

    def __init__(self, t):
        t = t._t if isinstance(t, DeviceFilterType) else t
        object.__setattr__(self, "_t", t)
        # what a device tensor must look like to pass untouched (checked per call, in this order)
        object.__setattr__(self, "_want", TORCH_DTYPES.get(getattr(t, "dtype", None)))
        object.__setattr__(self, "_ndim", getattr(t, "ndim", None))
        object.__setattr__(self, "_static", any(x is not None for x in getattr(t, "shape", ())))
        # host fast path: an ndarray that IS what ``TensorType.filter`` (tensor/type.py:153-156, then
        # the rank / alignment / static-shape checks :236-251) returns unchanged.  Built-in NumPy
        # dtypes are singletons, so the dtype test is an identity test; ``None`` disables the path
        # (a type that also checks finiteness, :253, or one without a NumPy dtype)
        npdt = getattr(t, "numpy_dtype", None)
        if getattr(t, "filter_checks_isfinite", False) or not isinstance(npdt, np.dtype) \
                or np.dtype(npdt.name) is not npdt:
            npdt = None
        object.__setattr__(self, "_np", npdt)

    def filter(self, value, strict=False, allow_downcast=None):
        if type(value) is np.ndarray:
            # (the 0-d ``mu`` / ``sigma`` / learning-rate arguments of every call: 0.3 us instead of
            # the 1.9 us of the general filter below, which reaches the same ``return data``)
            if value.dtype is self._np and value.ndim == self._ndim and not self._static \
                    and value.flags.aligned:
                return value
            return self._t.filter(value, strict=strict, allow_downcast=allow_downcast)
        if type(value) is torch.Tensor:
            if value.dtype is self._want and value.ndim == self._ndim and not self._static \
                    and value.device.type != "cpu":
                return value                     # the common case: right dtype and rank, on the device
        if isinstance(value, torch.Tensor):
            if value.device.type == "cpu":
                value = value.numpy()
            else:
                return filter_device_value(self._t, value, strict=strict,
                                           allow_downcast=allow_downcast)
        return self._t.filter(value, strict=strict, allow_downcast=allow_downcast)

    def is_valid_value(self, value, strict=True):
        if isinstance(value, torch.Tensor) and value.device.type != "cpu":
            try:
                filter_device_value(self._t, value, strict=strict)
                return True
            except TypeError:
                return False
        return self._t.is_valid_value(value, strict)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_t"), name)

    def __setattr__(self, name, value):
        if name in DeviceFilterType.__slots__:
            object.__setattr__(self, name, value)
        else:
            setattr(self._t, name, value)

    def __eq__(self, other):
        return self._t == (other._t if isinstance(other, DeviceFilterType) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._t)

    def __repr__(self):
        return repr(self._t)

    def __str__(self):
        return str(self._t)

    def __deepcopy__(self, memo):
        from copy import deepcopy
        return DeviceFilterType(deepcopy(self._t, memo))

    def __reduce__(self):
        return (DeviceFilterType, (self._t,))


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
