"""Device-array container and generated-kernel cache for the HIP executor.

PyTorch-ROCm is used strictly as plumbing: ``torch.empty`` is the device allocator (caching,
stream-aware), ``torch.cuda.current_stream()`` supplies the HIP stream, and tensors are how
results are handed back to the user.  All shape/stride arithmetic (views, broadcasting,
negative steps) is done here in element units and all compute goes through the C-ABI.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import threading

import numpy as np
import torch

from . import _lib
from . import knobs
from ._lib import DTYPE_CODES, check, lib

TORCH_DTYPES = {
    "bool": torch.bool, "int8": torch.int8, "int16": torch.int16, "int32": torch.int32,
    "int64": torch.int64, "uint8": torch.uint8, "uint16": torch.uint16,
    "uint32": torch.uint32, "uint64": torch.uint64, "float32": torch.float32,
    "float64": torch.float64,
}
_FROM_TORCH = {v: k for k, v in TORCH_DTYPES.items()}
ITEMSIZE = {k: np.dtype(k).itemsize for k in TORCH_DTYPES}


def contiguous_strides(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= max(int(s), 1)
    return tuple(reversed(st))


_DTYPE_NAMES = {}


def dtype_name(dt) -> str:
    """``np.dtype.name`` through a dict (NumPy 2 builds the string on every access: ~1 us, and the
    call path asks for it several times per input)."""
    try:
        return _DTYPE_NAMES[dt]
    except KeyError:
        name = _DTYPE_NAMES[dt] = dt.name
        return name


class DevArray:
    """A strided view of device memory: (buffer, element offset, shape, element strides)."""

    __slots__ = ("buf", "offset", "shape", "strides", "dtype")

    def __init__(self, buf, offset, shape, strides, dtype):
        self.buf = buf          # torch.Tensor owning the memory (kept alive by this view)
        self.offset = int(offset)
        self.shape = tuple(int(s) for s in shape)
        self.strides = tuple(int(s) for s in strides)
        self.dtype = dtype

    # -- construction -----------------------------------------------------------------
    @staticmethod
    def from_torch(t: torch.Tensor) -> "DevArray":
        if not t.is_cuda:
            raise TypeError("DevArray.from_torch needs a device (cuda/hip) tensor")
        return DevArray(t, 0, t.shape, t.stride(), _FROM_TORCH[t.dtype])

    @staticmethod
    def from_numpy(a: np.ndarray, device) -> "DevArray":
        a = np.asarray(a)
        name = dtype_name(a.dtype)
        if name not in TORCH_DTYPES:
            raise TypeError(f"dtype {name} is not supported on the HIP path")
        src = np.ascontiguousarray(a)
        if name in ("uint16", "uint32", "uint64"):
            t = torch.from_numpy(src.view(name[1:])).to(device).view(TORCH_DTYPES[name])
        else:
            t = torch.from_numpy(src).to(device)
        return DevArray(t, 0, a.shape, contiguous_strides(a.shape), name)

    # -- properties ---------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def itemsize(self):
        return ITEMSIZE[self.dtype]

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.offset * self.itemsize

    def is_contiguous(self):
        return self.size <= 1 or all(
            st == cs for s, st, cs in zip(self.shape, self.strides, contiguous_strides(self.shape))
            if s != 1)

    def view(self, shape, strides, offset=None):
        return DevArray(self.buf, self.offset if offset is None else offset, shape, strides,
                        self.dtype)

    def torch(self) -> torch.Tensor:
        """A torch view of this array (requires non-negative strides)."""
        if any(s < 0 for s in self.strides):
            raise ValueError("negative strides cannot be expressed as a torch view")
        base = self.buf
        if base.dtype != TORCH_DTYPES[self.dtype]:
            base = base.view(TORCH_DTYPES[self.dtype])
        return torch.as_strided(base, self.shape, self.strides,
                                base.storage_offset() + self.offset)

    def numpy(self) -> np.ndarray:
        t = self.torch()
        if self.dtype in ("uint16", "uint32", "uint64"):
            return t.contiguous().view(TORCH_DTYPES[self.dtype[1:]]).cpu().numpy().view(self.dtype)
        return t.cpu().numpy()

    def __repr__(self):
        return f"DevArray({self.dtype}, shape={self.shape}, strides={self.strides}, off={self.offset})"


# ---------------------------------------------------------------------------------------
# generated-kernel cache: source hash -> code object (disk, in-tree) -> loaded module
# reference analogue: link/c/cmodule.py:618 ModuleCache keyed by get_module_hash (:419)
# ---------------------------------------------------------------------------------------
CACHE_DIR = knobs.get("KCACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_kcache")
_lock = threading.Lock()
_loaded = {}   # sha -> (module handle, {name: fn handle})


def compile_cached(source: str) -> bytes:
    sha = hashlib.sha256((source + "|gfx950|-O3").encode()).hexdigest()[:32]
    path = os.path.join(CACHE_DIR, sha + ".hsaco")
    if os.path.exists(path):
        with open(path, "rb") as f:
            return f.read()
    code = _lib.compile_source(source, name=sha + ".hip")
    try:
        os.makedirs(CACHE_DIR, exist_ok=True)
        tmp = path + ".%d.tmp" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(code)
        os.replace(tmp, path)
    except OSError:
        pass
    return code


def load_kernels(source: str, names):
    """Compile (or fetch) and load a generated module; returns the fn handles for ``names``."""
    sha = hashlib.sha256(source.encode()).hexdigest()
    with _lock:
        ent = _loaded.get(sha)
        if ent is None:
            code = compile_cached(source)
            mod = C.c_void_p()
            keep = C.create_string_buffer(code, len(code))
            check(lib.ahip_module_load(keep, len(code), C.byref(mod)))
            ent = (mod, {}, keep)
            _loaded[sha] = ent
        fns = []
        for n in names:
            if n not in ent[1]:
                fn = C.c_void_p()
                check(lib.ahip_module_get_function(ent[0], n.encode(), C.byref(fn)))
                ent[1][n] = fn
            fns.append(ent[1][n])
        return fns


def dtype_code(dt):
    return DTYPE_CODES[dt]


def current_stream_handle():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
