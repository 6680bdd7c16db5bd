"""ctypes binding of the C-ABI ``libaesara_hip.so`` (include/aesara_hip.h).

The product path has NO CPU fallback: if the shared library is missing or does not export the
declared ABI, importing this module raises ``HipLibraryMissing`` — loudly.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64 and must load it BEFORE libaesara_hip.so pulls in the
# system copy: the process then has ONE HIP runtime (the dynamic loader resolves our dependency
# to the already loaded SONAME).  Loaded the other way round there are two runtimes, and a static
# kernel of this library launched on one of torch's streams fails with "no ROCm-capable device".
import torch  # noqa: F401,E402

from . import knobs  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = knobs.get("LIB") or os.path.join(_HERE, "libaesara_hip.so")

AHIP_MAXD = 6
AHIP_MAXOPS = 32
ABI_VERSION = 1

DTYPE_CODES = {
    "bool": 0, "int8": 1, "int16": 2, "int32": 3, "int64": 4,
    "uint8": 5, "uint16": 6, "uint32": 7, "uint64": 8,
    "float32": 9, "float64": 10,
}

AHIP_OK, AHIP_EINVAL, AHIP_EHIP, AHIP_ECOMPILE, AHIP_EINDEX, AHIP_ENOSUP = 0, -1, -2, -3, -4, -5


class HipLibraryMissing(ImportError):
    pass


class HipError(RuntimeError):
    pass


class DeviceInfo(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("cu_count", C.c_int32), ("wavefront_size", C.c_int32),
        ("max_threads_per_block", C.c_int32), ("total_mem", C.c_int64),
        ("lds_per_block", C.c_int64), ("clock_khz", C.c_int32), ("l2_bytes", C.c_int32),
        ("arch", C.c_char * 64), ("name", C.c_char * 128),
    ]


AHIP_MAXDOTS = 8
AHIP_GV_MAXOPS = 16


AHIP_GV_MAXXIN = 4


class GvArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int64),
        ("A", C.c_void_p * AHIP_MAXDOTS), ("a_rs", C.c_int64 * AHIP_MAXDOTS),
        ("a_cs", C.c_int64 * AHIP_MAXDOTS), ("K", C.c_int64 * AHIP_MAXDOTS),
        ("x", C.c_void_p * AHIP_MAXDOTS), ("incx", C.c_int64 * AHIP_MAXDOTS),
        ("ptr", C.c_void_p * AHIP_GV_MAXOPS), ("stride", C.c_int64 * AHIP_GV_MAXOPS),
        ("ndots", C.c_int32), ("nops", C.c_int32),
        ("xin", (C.c_void_p * AHIP_GV_MAXXIN) * AHIP_MAXDOTS), ("xout", C.c_void_p * AHIP_MAXDOTS),
    ]


AHIP_RP_MAXOPS = 16


class RpArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("K", C.c_int64), ("X", C.c_void_p), ("x_rs", C.c_int64),
        ("w", C.c_void_p),
        ("ptr", C.c_void_p * AHIP_RP_MAXOPS), ("stride", C.c_int64 * AHIP_RP_MAXOPS),
        ("col_ws", C.c_void_p), ("red_ws", C.c_void_p),
        ("nops", C.c_int32), ("nred", C.c_int32),
    ]


AHIP_GE_MAXDOTS = 3
AHIP_GE_MAXOPS = 12


class GeArgs(C.Structure):
    _fields_ = [("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64 * AHIP_GE_MAXDOTS),
                ("A", C.c_void_p * AHIP_GE_MAXDOTS), ("a_rs", C.c_int64 * AHIP_GE_MAXDOTS),
                ("B", C.c_void_p * AHIP_GE_MAXDOTS), ("b_rs", C.c_int64 * AHIP_GE_MAXDOTS),
                ("b_cs", C.c_int64 * AHIP_GE_MAXDOTS),
                ("ptr", C.c_void_p * AHIP_GE_MAXOPS), ("rs", C.c_int64 * AHIP_GE_MAXOPS),
                ("cs", C.c_int64 * AHIP_GE_MAXOPS)]


AHIP_RC_MAXOPS = 16
AHIP_RC_MAXLEAD = 4


class RcArgs(C.Structure):
    _fields_ = [("N", C.c_int64), ("K", C.c_int64), ("lshape", C.c_int64 * AHIP_RC_MAXLEAD),
                ("ptr", C.c_void_p * AHIP_RC_MAXOPS),
                ("ls", (C.c_int64 * AHIP_RC_MAXLEAD) * AHIP_RC_MAXOPS)]


vp, i64, i32, u32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_size_t
p_i64 = C.POINTER(C.c_int64)
p_vp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  This table IS the Python view of include/aesara_hip.h;
# tests/test_abi.py checks that every symbol declared in the header is listed and exported.
SIGNATURES = {
    "ahip_abi_version": (i32, []),
    "ahip_init": (i32, [i32]),
    "ahip_last_error": (C.c_char_p, []),
    "ahip_get_device_info": (i32, [C.POINTER(DeviceInfo)]),
    "ahip_stream_synchronize": (i32, [vp]),
    "ahip_compile": (i32, [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), i32, p_vp,
                           C.POINTER(sz)]),
    "ahip_free_code": (i32, [vp]),
    "ahip_module_load": (i32, [vp, sz, p_vp]),
    "ahip_module_get_function": (i32, [vp, C.c_char_p, p_vp]),
    "ahip_module_unload": (i32, [vp]),
    "ahip_launch": (i32, [vp, u32, u32, u32, u32, u32, u32, u32, vp, sz, vp]),
    "ahip_launch_p": (i32, [vp, u32, u32, u32, u32, u32, u32, u32, vp, sz, C.POINTER(C.c_uint16), i32,
                            i32, vp]),
    "ahip_occupancy": (i32, [vp, i32, sz, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ahip_elemwise": (i32, [vp, i32, p_i64, i32, p_vp, p_i64, i32, i32, vp]),
    "ahip_elemwise_wg": (i32, [vp, i32, p_i64, i32, p_vp, p_i64, i32, i32, i32, vp]),
    "ahip_elemwise_tiled": (i32, [vp, i32, p_i64, i32, p_vp, p_i64, i32, i32, vp, vp, sz, vp]),
    "ahip_reduce_ws_bytes": (sz, []),
    "ahip_reduce_partials_bytes": (sz, []),
    "ahip_elemwise_reduce_all": (i32, [vp, i32, p_i64, i32, p_vp, p_i64, i32, i32, vp, vp, sz,
                                       vp]),
    "ahip_elemwise_reduce_all_multi": (i32, [vp, i32, i32, p_vp, p_i64, i32, i32, p_vp, vp, sz, vp]),
    "ahip_set_param": (i32, [C.c_char_p, i64]),
    "ahip_elemwise_reduce_axis": (i32, [vp, i32, i32, i32, p_i64, i32, p_vp, p_i64, i32, vp,
                                        i32, i32, i32, vp]),
    "ahip_gemm": (i32, [i32, i64, i64, i64, vp, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64,
                        vp, i64, i64, vp]),
    "ahip_gemm_ws_bytes": (sz, [i32, i64, i64, i64, i64]),
    "ahip_gemm_splitk": (i32, [i32, i64, i64, i64, vp, vp, i64, i64, vp, i64, i64, vp, vp, i64, i64,
                               vp, i64, i64, vp, sz, vp]),
    "ahip_gemm_batched": (i32, [i32, i64, i64, i64, i64, vp, vp, i64, i64, i64, vp, i64, i64,
                                i64, vp, vp, i64, i64, i64, vp, i64, i64, i64, vp]),
    "ahip_igemm_batched": (i32, [i32, i64, i64, i64, i64, vp, i64, i64, i64, vp, i64, i64, i64, vp,
                                 i64, i64, i64, vp]),
    "ahip_gemv_ws_bytes": (sz, [i32, i64, i64]),
    "ahip_gemv": (i32, [i32, i64, i64, vp, vp, i64, i64, vp, i64, vp, vp, i64, vp, i64, vp, sz,
                        vp]),
    "ahip_gemv_epilogue": (i32, [vp, C.POINTER(GvArgs), i32, vp]),
    "ahip_rowpass_grid": (i32, [i64, i32, i32]),
    "ahip_rowpass": (i32, [vp, C.POINTER(RpArgs), i32, i32, sz, vp]),
    "ahip_rowchain": (i32, [vp, C.POINTER(RcArgs), i32, i32, vp]),
    "ahip_gemm_epilogue": (i32, [vp, C.POINTER(GeArgs), i32, i32, vp]),
    "ahip_ger": (i32, [i32, i64, i64, vp, vp, i64, vp, i64, vp, i64, i64, vp, i64, i64, vp]),
    "ahip_copy_strided": (i32, [i32, i32, p_i64, vp, p_i64, vp, p_i64, i32, vp]),
    "ahip_fill": (i32, [i32, vp, vp, i64, vp]),
    "ahip_take_rows": (i32, [i32, vp, i64, i64, i64, vp, i32, i64, i64, vp, i64, vp, vp]),
    "ahip_scatter_rows": (i32, [i32, vp, i64, i64, i64, vp, i32, i64, i64, vp, i64, i32, vp,
                                vp]),
    "ahip_scatter_add_ws_bytes": (sz, [i64, i64]),
    "ahip_scatter_add_rows_ordered": (i32, [i32, vp, i64, i64, i64, vp, i32, i64, i64, vp, i64, vp,
                                            sz, vp, vp]),
    "ahip_sort_max_row": (i32, [i32]),
    "ahip_sort_rows": (i32, [i32, vp, i64, i64, i64, i64, vp, vp, vp]),
    "ahip_sort_large_ws_bytes": (sz, [i32, i64, i64]),
    "ahip_sort_rows_large": (i32, [i32, vp, i64, i64, i64, i64, vp, vp, vp, sz, vp]),
    "ahip_nonzero_write": (i32, [vp, i64, i32, p_i64, p_vp, vp]),
    "ahip_argmax_ws_bytes": (sz, [i32, i64, i64, i64, i64]),
    "ahip_argmax_rows": (i32, [i32, vp, i64, i64, i64, i64, vp, vp, sz, vp]),
    "ahip_arange": (i32, [i32, vp, vp, i64, vp, vp]),
    "ahip_cumulative_ws_bytes": (sz, [i32, i64, i64, i64]),
    "ahip_cumulative": (i32, [i32, i32, vp, i64, i64, i64, i64, i64, i64, vp, vp, sz, vp]),
    "ahip_linearize_indices": (i32, [i32, p_vp, C.POINTER(C.c_int), p_i64, p_i64, p_i64, i64, vp,
                                     vp, vp]),
    "ahip_searchsorted": (i32, [i32, vp, i64, i64, vp, i64, i32, vp, vp, vp]),
    "ahip_list_begin": (i32, []),
    "ahip_list_end": (i32, [p_vp]),
    "ahip_list_length": (i32, [vp]),
    "ahip_list_run": (i32, [vp, vp]),
    "ahip_list_bind_bases": (i32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i32]),
    "ahip_list_run_rebased": (i32, [vp, C.POINTER(C.c_uint64), i32, vp]),
    "ahip_list_destroy": (i32, [vp]),
    "ahip_graph_begin": (i32, [vp]),
    "ahip_graph_end": (i32, [vp, p_vp]),
    "ahip_graph_launch": (i32, [vp, vp]),
    "ahip_graph_destroy": (i32, [vp]),
    "ahip_comm_set_library": (i32, [C.c_char_p]),
    "ahip_comm_unique_id": (i32, [vp, sz]),
    "ahip_comm_init_rank": (i32, [vp, i32, i32, p_vp]),
    "ahip_comm_size": (i32, [vp]),
    "ahip_comm_rank": (i32, [vp]),
    "ahip_allreduce": (i32, [vp, i32, i32, vp, vp, i64, vp]),
    "ahip_comm_destroy": (i32, [vp]),
    "ahip_event_create": (i32, [p_vp]),
    "ahip_event_record": (i32, [vp, vp]),
    "ahip_event_elapsed_ms": (i32, [vp, vp, C.POINTER(C.c_float)]),
    "ahip_event_query": (i32, [vp]),
    "ahip_event_destroy": (i32, [vp]),
    "ahip_comm_abort": (i32, [vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C aesara_amd/csrc`.  There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HipLibraryMissing(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryMissing(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ahip_abi_version() != ABI_VERSION:
        raise HipLibraryMissing("libaesara_hip.so ABI version mismatch; rebuild")
    return lib


lib = _load()


_PTR_OFFS = {}


def ptr_offsets(struct_cls):
    """(ctypes uint16 array, n): byte offsets of the device pointers (``c_void_p`` fields and
    arrays of them) of a kernel-argument Structure — the pointer map ``ahip_launch_p`` takes, so
    a recorded launch is only ever re-pointed at words that ARE pointers."""
    ent = _PTR_OFFS.get(struct_cls)
    if ent is None:
        offs = []
        for name, typ in struct_cls._fields_:
            base = getattr(struct_cls, name).offset
            if typ is C.c_void_p:
                offs.append(base)
            elif isinstance(typ, type) and issubclass(typ, C.Array):
                t, n = typ, 1
                while isinstance(t, type) and issubclass(t, C.Array):
                    n *= t._length_
                    t = t._type_
                if t is C.c_void_p:
                    offs.extend(base + 8 * i for i in range(n))
        ent = _PTR_OFFS[struct_cls] = ((C.c_uint16 * max(len(offs), 1))(*offs), len(offs))
    return ent


COMM_ID_BYTES = 128
RED_OPS = {"add": 0, "mul": 1, "maximum": 2, "minimum": 3}


def check(rc):
    """Translate a C-ABI return code into the reference's exception types."""
    if rc == 0:
        return
    msg = (lib.ahip_last_error() or b"").decode("utf-8", "replace")
    if rc == AHIP_EINVAL:
        raise ValueError(msg)
    if rc == AHIP_EINDEX:
        raise IndexError(msg)
    if rc == AHIP_ENOSUP:
        raise NotImplementedError(msg)
    raise HipError(f"[{rc}] {msg}")


def compile_source(source: str, name: str = "k.hip", options=("-O3",)) -> bytes:
    """HIP source -> gfx950 code object (hiprtc; works without a GPU)."""
    opts = (C.c_char_p * len(options))(*[o.encode() for o in options])
    code = C.c_void_p()
    size = C.c_size_t()
    check(lib.ahip_compile(source.encode(), name.encode(), opts, len(options), C.byref(code),
                           C.byref(size)))
    try:
        return C.string_at(code, size.value)
    finally:
        lib.ahip_free_code(code)


def device_info() -> DeviceInfo:
    info = DeviceInfo()
    check(lib.ahip_get_device_info(C.byref(info)))
    return info
