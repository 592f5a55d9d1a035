"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/m3d_hip.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "m3d_hip.h")

_CTYPE = {
    "int": ctypes.c_int32, "int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "int64_t": ctypes.c_int64,
    "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64,
}


def _prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"#.*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|size_t)\s+(m3d_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, params)
    return protos


def _ctype_of(param: str):
    if "*" in param:
        return ctypes.c_void_p
    t = param.replace("const", " ").split()
    return _CTYPE[t[0]]


def test_library_builds_and_exports_every_declared_symbol():
    from myria3d_amd import _lib

    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    protos = _prototypes()
    assert len(protos) >= 20
    for name in protos:
        assert hasattr(handle, name), f"{name} declared in m3d_hip.h but not exported by libm3d_hip.so"
    handle.m3d_abi_version.restype = ctypes.c_int32
    assert handle.m3d_abi_version() == _lib.ABI_VERSION


def test_ctypes_signatures_mirror_the_header():
    from myria3d_amd import _lib

    protos = _prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (ret, params) in protos.items():
        res, args = _lib.SIGNATURES[name]
        assert len(args) == len(params), f"{name}: {len(args)} ctypes args vs {len(params)} in the header"
        for i, (a, p) in enumerate(zip(args, params)):
            assert a is _ctype_of(p), f"{name} arg {i}: {a} vs `{p}`"
        assert res is (ctypes.c_size_t if ret == "size_t" else ctypes.c_int32)


def test_workspace_query_runs_without_gpu():
    from myria3d_amd import _lib

    n = _lib.lib().m3d_knn_workspace_bytes(12800 * 16, 16)
    assert n >= 12800 * 16 * 16 + 16 * 4097 * 4


def test_invalid_arguments_return_error_codes_not_crashes():
    from myria3d_amd import _lib

    h = _lib.lib()
    assert h.m3d_knn_query(None, None, 10, 1, None, 3, None, None, 10, 16, 0, None, None, None) == -1
    assert h.m3d_gemm_f32(None, 0, 0, None, 4, None, 0, 0, None, 0, 0, 8, 8, None, None, None, 0, 0.2, None, 0,
                          None, 0, 0, 1, None) == -1
    assert h.m3d_lfa_fwd(None, None, None, 10, 16, 8, None, None, None, 0.2, None, 0, None) == -1
    with pytest.raises(_lib.M3DError):
        _lib.call("m3d_bn_apply", None, None, None, None, None, None, 1, 0.2, None, 10, 8, None)
