"""ctypes binding of ``libm3d_hip.so`` (the C ABI declared in ``include/m3d_hip.h``).

The library is built in-tree by ``myria3d_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is no CPU
fallback: if the shared object is missing or a call returns a non-zero status the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M3D_LIB") or os.path.join(_HERE, "libm3d_hip.so")  # M3D_LIB: tuning-sweep variants
CSRC_DIR = os.path.join(_HERE, "csrc")

_p, _i32, _i64, _f32, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32

# name -> (restype, argtypes): must mirror include/m3d_hip.h exactly (tests/test_abi.py checks the symbol list)
SIGNATURES = {
    "m3d_abi_version": (_i32, []),
    "m3d_stream_capture_id": (_i32, [_p, _p]),
    "m3d_copy_many": (_i32, [_p, _p, _p, _i32, _p]),
    "m3d_knn_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "m3d_knn_build": (_i32, [_p, _i32, _p, _i32, _i64, _p, _p]),
    "m3d_knn_build_map": (_i32, [_p, _i32, _p, _i32, _i64, _p, _p, _p, _p]),
    "m3d_knn_workspace_offset": (C.c_size_t, [_i64, _i32, _i32]),
    "m3d_knn_query": (_i32, [_p, _p, _i64, _i32, _p, _i32, _p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "m3d_gemm_stat_parts": (_i32, [_i64, _i32, _i32]),
    "m3d_gemm_f32": (_i32, [_p, _i64, _i32, _p, _i32, _p, _i64, _i32, _p, _i64, _i32, _i64, _i32, _p, _p, _p, _i32,
                            _f32, _p, _i32, _p, _i64, _i32, _i32, _p]),
    "m3d_gemm_pair_f32": (_i32, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p, _i32, _p, _p, _p, _i32, _p]),
    "m3d_linear_wgrad_workspace_bytes": (C.c_size_t, [_i64, _i32, _i32]),
    "m3d_linear_wgrad_f32": (_i32, [_p, _i64, _p, _i64, _p, _i32, _p, _i64, _i32, _i64, _i32, _p, _i64, _i32, _p, _p]),
    "m3d_linear_wgrad_batch": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p]),
    "m3d_colsum_f32": (_i32, [_p, _i64, _i64, _i32, _p, _p]),
    "m3d_colsum_bf16": (_i32, [_p, _i64, _i64, _i32, _p, _p]),
    "m3d_bn_finalize": (_i32, [_p, _i32, _i64, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "m3d_bn_fold_eval": (_i32, [_p, _p, _p, _p, _f32, _p, _p, _i32, _p]),
    "m3d_bn_apply": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _f32, _p, _i64, _i32, _p]),
    "m3d_bn_stats_apply": (_i32, [_p, _i32, _i64, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                  _p, _p, _p, _p, _i32, _f32, _p, _i64, _i32, _p, _p]),
    "m3d_bn_bwd_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "m3d_bn_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _f32, _i64, _i32, _p, _p, _p, _p, _p,
                          _p, _p, _i32, _p, _p]),
    "m3d_bn_dgrad_f32": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _f32, _p, _i32, _i64, _i32, _p, _i64, _i32, _p, _i64, _p, _p,
                                _p, _i32, _i32, _p, _i64, _p, _p]),
    "m3d_gather_rows": (_i32, [_p, _i64, _p, _p, _i64, _i32, _p]),
    "m3d_gather_rows_bf16": (_i32, [_p, _i64, _p, _p, _i64, _i32, _p]),
    "m3d_convert_f32_bf16": (_i32, [_p, _p, _i64, _p]),
    "m3d_scatter_add_rows": (_i32, [_p, _p, _p, _i64, _i64, _i32, _i32, _p]),
    "m3d_csr_invert_batch": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _p]),
    "m3d_gather_sum_rows": (_i32, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _p]),
    "m3d_pad_pos": (_i32, [_p, _i32, _p, _i64, _p]),
    "m3d_decimation_indices": (_i32, [_p, _p, _i32, _p, _u32, _p, _i64, _p]),
    "m3d_decimate_level": (_i32, [_p, _p, _i32, _p, _u32, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "m3d_lfa_moments": (_i32, [_p, _p, _i64, _i32, _p, _p]),
    "m3d_lfa_moments_batch": (_i32, [_i32, _p, _p, _p, _i32, _p, _i64, _p]),
    "m3d_knn_query_batch": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "m3d_lfa_enc_finalize": (_i32, [_p, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "m3d_lfa_pack_att": (_i32, [_p, _i32, _p, _p, _p]),
    "m3d_lfa_prepare": (_i32, [_p, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _i32, _p, _i32, _p, _p, _i32,
                               _p]),
    "m3d_lfa_prepare_batch": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                     _p]),
    "m3d_gemm_bn_on_load_f32": (_i32, [_p, _p, _i32, _p, _i64, _i64, _i32, _p, _p, _i32, _p, _i64, _p]),
    "m3d_lfa_fwd_bf16": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p, _f32, _p, _i32, _p]),
    "m3d_lfa_pack_att_bf16": (_i32, [_p, _i32, _p, _p, _p]),
    "m3d_lfa_fwd": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p, _f32, _p, _i32, _p]),
    "m3d_lfa_bwd_workspace_bytes": (C.c_size_t, [_i64, _i32, _i32]),
    "m3d_lfa_bwd_edge_rows_ok": (_i32, [_i64, _i32, _i32, C.c_float]),
    "m3d_knn_reverse_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "m3d_knn_reverse": (_i32, [_p, _i64, _i32, _p, _p, _p, _p, _p]),
    "m3d_lfa_bwd_edge_rows": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p, _p, _f32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    "m3d_lfa_bwd": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p, _p, _f32, _p, _p, _p, _i32, _p, _p, _p]),
    "m3d_lfa_bwd_bf16": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p, _p, _f32, _p, _p, _p, _i32, _p, _p, _p]),
    "m3d_lfa_edge_features": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _f32, _p, _p]),
    "m3d_lfa_edge_softmax_fwd": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p]),
    "m3d_lfa_edge_softmax_bwd": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "m3d_lfa_edge_features_bwd": (_i32, [_p, _p, _p, _i64, _i32, _i32, _p, _p, _f32, _p, _p, _p]),
    "m3d_lfa_enc_bwd_finalize": (_i32, [_p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p]),
    "m3d_lfa_enc_bwd_finalize_batch": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "m3d_lfa_bwd_reduce_batch": (_i32, [_i32, _p, _p, _p, _p, _p, _p, _p]),
    "m3d_idw_interpolate_fwd": (_i32, [_p, _i64, _p, _p, _i64, _i32, _i32, _p, _p]),
    "m3d_predict_reduce": (_i32, [_p, _i64, _p, _i64, _i32, _p, _i64, _p, _p, _p]),
    "m3d_grid_sampling_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "m3d_grid_sampling": (_i32, [_p, _i32, _p, _i64, _i32, _p, _p, _i32, _i64, _f32, _p, _p, _p, _p, _p, _p]),
    "m3d_grid_sampling_status": (_i32, [_p, _i64, _i32, _p, _p]),
    "m3d_tile_select_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "m3d_tile_select": (_i32, [_p, _i32, _i64, _p, _i32, C.c_double, C.c_double, C.c_double, _p, _i32, _p, _p, _p]),
    "m3d_tile_normalize": (_i32, [_p, _i32, _p, _i64, _i32, _i32, _p, _i32, _i64, _i32, _i32, _f32, _f32, _p, _p]),
    "m3d_fps_sorted": (_i32, [_p, _i64, _p, _p, _i32, _i64, _p, _p, _p]),
    "m3d_fps": (_i32, [_p, _p, _p, _i32, _i64, _p, _p, _p]),
    "m3d_sa_group": (_i32, [_p, _i64, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p, _p]),
    "m3d_sa_group_bwd": (_i32, [_p, _i64, _p, _i64, _i32, _p, _i64, _p]),
    "m3d_seg_max": (_i32, [_p, _i64, _p, _i64, _i32, _p, _p, _p]),
    "m3d_seg_max_bwd": (_i32, [_p, _p, _p, _p, _i64, _i32, _p, _p]),
    "m3d_ce_loss_fwd": (_i32, [_p, _i64, _p, _i64, _i32, _i64, _p, _p, _p, _i32, _p]),
    "m3d_ce_loss_bwd": (_i32, [_p, _i64, _p, _i64, _i32, _i64, _p, _p, _p, _p, _p]),
    "m3d_adam_step": (_i32, [_p, _p, _p, _p, _p, _p, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _i64, _p]),
    "m3d_zero_bump": (_i32, [_p, _i64, _p, _i32, _p]),
    "m3d_dropout": (_i32, [_p, _p, _i64, _f32, _p, C.c_uint64, _p]),
}

class M3DDropout(C.Structure):
    """``M3DDropout`` of include/m3d_hip.h (read by the library on the host, at call time)."""
    _fields_ = [("counter", C.c_void_p), ("seed", C.c_uint64), ("p", C.c_float), ("rows", C.c_void_p),
                ("snapshot", C.c_void_p)]


class M3DBnOnLoad(C.Structure):
    """``M3DBnOnLoad`` of include/m3d_hip.h (read by the library on the host, at call time)."""
    _fields_ = [("slots", C.c_void_p), ("nslots", C.c_int32), ("count", C.c_int64), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p),
                ("invstd", C.c_void_p), ("act", C.c_int32), ("slope", C.c_float), ("y", C.c_void_p)]


ABI_VERSION = 18  # M3D_ABI_VERSION in include/m3d_hip.h

_ERRORS = {-1: "invalid argument", -2: "unsupported shape", -3: "kernel launch failure"}


class M3DError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``libm3d_hip.so`` (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise M3DError("building libm3d_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise M3DError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C myria3d_amd/csrc`). myria3d_amd has no CPU/eager fallback by design."
            )
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if handle.m3d_abi_version() != ABI_VERSION:
            raise M3DError("libm3d_hip.so ABI version mismatch; rebuild")
        _lib = handle
    return _lib


_FN: dict = {}  # name -> bound ctypes function (skips the library / attribute look-up on the hot path)


def call(name: str, *args):
    """Invoke an int-returning entry point and raise :class:`M3DError` on a non-zero status."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise M3DError(f"{name} failed: {_ERRORS.get(rc, rc)}")
