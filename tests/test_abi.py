"""The C-ABI library loads and exports every symbol include/aesara_hip.h declares, with the
signatures the ctypes table binds (no compute calls here: no GPU in this tier)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aesara_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = re.findall(r"^\s*(?:int|size_t|const char\s*\*)\s+(ahip_\w+)\s*\(([^;]*?)\)\s*;", src,
                       flags=re.M | re.S)
    return {name: [a.strip() for a in args.split(",")] if args.strip() != "void" else []
            for name, args in decls}


def test_header_declares_entry_points():
    fns = declared_functions()
    assert len(fns) >= 30
    for must in ("ahip_elemwise", "ahip_elemwise_reduce_all", "ahip_gemm", "ahip_gemv",
                 "ahip_copy_strided", "ahip_take_rows", "ahip_graph_launch", "ahip_compile"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from aesara_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_table_matches_header():
    from aesara_amd import _lib
    fns = declared_functions()
    assert set(fns) == set(_lib.SIGNATURES), set(fns) ^ set(_lib.SIGNATURES)
    for name, args in fns.items():
        assert len(args) == len(_lib.SIGNATURES[name][1]), name


def test_abi_version_and_struct_layout():
    from aesara_amd import _lib
    assert _lib.lib.ahip_abi_version() == _lib.ABI_VERSION
    # ahip_ew_args layout the generated kernels rely on
    size = 8 + 8 * _lib.AHIP_MAXD + 8 * _lib.AHIP_MAXOPS * _lib.AHIP_MAXD + 8 * _lib.AHIP_MAXOPS \
        + 8 * 4 + 4 * 2
    assert size == 1888 and size < 4096  # kernarg segment limit
    assert _lib.lib.ahip_reduce_ws_bytes() >= 4096 * 8


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import sys
    monkeypatch.setenv("AESARA_HIP_LIB", str(tmp_path / "nope.so"))
    saved = sys.modules.pop("aesara_amd._lib")
    try:
        try:
            importlib.import_module("aesara_amd._lib")
        except ImportError as e:
            assert "no CPU fallback" in str(e) or "not found" in str(e)
        else:
            raise AssertionError("import must fail without the HIP library")
    finally:
        sys.modules["aesara_amd._lib"] = saved


def test_hiprtc_cross_compiles_without_gpu():
    from aesara_amd import _lib
    code = _lib.compile_source(
        'extern "C" __global__ void k(double* o, long n){ long i = blockIdx.x * 256 + '
        'threadIdx.x; if (i < n) o[i] = exp(o[i]); }')
    assert code[:4] == b"\x7fELF"


def test_compile_error_is_reported():
    import pytest
    from aesara_amd import _lib
    with pytest.raises(_lib.HipError) as ei:
        _lib.compile_source("this is not HIP")
    assert "error" in str(ei.value)


def test_pointer_maps_of_the_kernel_argument_structs():
    """``ptr_offsets``: what ``ahip_launch_p`` is told about a kernarg block — every ``c_void_p``
    field / array element, nothing else (scalars are never candidates for rebinding)."""
    from aesara_amd import _lib
    from aesara_amd.scan_persist import SpArgs, SP_MAXMAT, SP_MAXSEQ, SP_MAXNSQ, SP_MAXOUT
    from aesara_amd.scan_persist_mat import SmArgs
    offs, n = _lib.ptr_offsets(SpArgs)
    assert n == SP_MAXMAT + SP_MAXSEQ + SP_MAXNSQ + SP_MAXOUT + 2
    got = list(offs)[:n]
    assert got == sorted(got) and all(o % 8 == 0 and o + 8 <= C.sizeof(SpArgs) for o in got)
    assert SpArgs.T.offset not in got and SpArgs.xch.offset in got and SpArgs.ctl.offset in got
    assert SpArgs.mat.offset in got and SpArgs.mat_rs.offset not in got
    offs2, n2 = _lib.ptr_offsets(SmArgs)
    assert n2 > 0 and SmArgs.out.offset in list(offs2)[:n2] and SmArgs.out_ts.offset not in list(offs2)[:n2]


def test_collective_entry_points_validate_arguments_without_a_gpu():
    """No RCCL call is made: only the argument checks of the shim (no GPU in this tier)."""
    from aesara_amd import _lib
    lib = _lib.lib
    assert lib.ahip_comm_size(None) == -1 and lib.ahip_comm_rank(None) == -1
    assert lib.ahip_allreduce(None, 10, 0, None, None, 4, None) == _lib.AHIP_EINVAL
    assert b"null communicator" in lib.ahip_last_error()
    buf = C.create_string_buffer(16)
    assert lib.ahip_comm_unique_id(buf, 16) == _lib.AHIP_EINVAL          # id buffer too small
    assert lib.ahip_comm_init_rank(buf, 2, 5, C.byref(C.c_void_p())) == _lib.AHIP_EINVAL
    assert lib.ahip_comm_destroy(None) == 0
