"""ctypes binding of libqe_hip.so -- the same symbols a JVM host would bind
through Panama/JNI.  There is no CPU fallback: if the library
is missing, or no HIP device is present, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
import subprocess
from typing import List, Optional, Sequence

import numpy as np

from .datatypes import DataType

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libqe_hip.so")

OK = 0
ERR_NAMES = {1: "INVALID_ARG", 2: "PROGRAM", 3: "HIP", 4: "OOM", 5: "UNSUPPORTED", 6: "INTERNAL", 7: "COMM"}
DEVICE_NONE = -1
EXEC_FUSED, EXEC_PER_NODE = 0, 1
CMP_TOTAL_ORDER, CMP_IEEE = 0, 1
GEN_I64_MOD, GEN_I32_MOD, GEN_F64_UNIT, GEN_F64_MOD, GEN_F64_STEP, GEN_F64_PRICE, GEN_DICT_MOD, GEN_I64_ROWID = range(8)
AGG_MIN, AGG_MAX, AGG_SUM, AGG_COUNT, AGG_AVG = range(5)
COMM_ID_BYTES = 128
FORM_RING, FORM_TWO_PASS, FORM_DENSE, FORM_PER_NODE, FORM_NO_FILTER, FORM_LOCAL = range(6)
FORM_GROUPBY_DENSE, FORM_GROUPBY_HASHED, FORM_GROUPBY_HASH_PARTITIONED = 8, 9, 10


class QeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[QE_ERR_{ERR_NAMES.get(code, code)}] {msg}")
        self.code = code


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("exec_mode", C.c_int32), ("cmp_semantics", C.c_int32),
                ("profile", C.c_int32), ("result_capacity_rows", C.c_int64), ("jit_cache_dir", C.c_char_p),
                ("tuning", C.c_int32 * 8)]


class ColDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("reserved", C.c_int32), ("data", C.c_void_p), ("validity", C.c_void_p),
                ("dict", C.c_void_p)]


class GenSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("col_id", C.c_int32), ("modulus", C.c_uint64), ("offset", C.c_int64),
                ("step", C.c_double), ("aux_col_id", C.c_int32), ("null_pct", C.c_int32), ("dict", C.c_void_p)]


class ColView(C.Structure):
    _fields_ = [("type", C.c_int32), ("nullable", C.c_int32), ("data", C.c_void_p), ("validity", C.c_void_p),
                ("count", C.c_int64), ("dict", C.c_void_p)]


# every symbol include/qe_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("qe_abi_version", C.c_int32, []),
    ("qe_last_error", C.c_char_p, [_P]),
    ("qe_ctx_create", C.c_int32, [C.c_int32, C.POINTER(Options), C.POINTER(_P)]),
    ("qe_ctx_destroy", None, [_P]),
    ("qe_ctx_set_exec_mode", C.c_int32, [_P, C.c_int32]),
    ("qe_ctx_set_cmp_semantics", C.c_int32, [_P, C.c_int32]),
    ("qe_ctx_kernel_time", C.c_int32, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("qe_ctx_reset_kernel_time", C.c_int32, [_P]),
    ("qe_ctx_last_form", C.c_int32, [_P]),
    ("qe_ctx_synchronize", C.c_int32, [_P]),
    ("qe_ctx_trim", C.c_int32, [_P]),
    ("qe_dict_create", C.c_int32, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P)]),
    ("qe_dict_size", C.c_int32, [_P]),
    ("qe_dict_entry", C.c_char_p, [_P, C.c_int32]),
    ("qe_dict_free", None, [_P, _P]),
    ("qe_batch_create", C.c_int32, [_P, C.c_int64, C.c_int32, C.POINTER(ColDesc), C.POINTER(_P)]),
    ("qe_batch_wrap_device", C.c_int32, [_P, C.c_int64, C.c_int32, C.POINTER(ColDesc), C.POINTER(_P)]),
    ("qe_batch_describe", C.c_int32, [_P, C.c_int64, C.c_int32, C.POINTER(ColDesc), C.POINTER(_P)]),
    ("qe_batch_generate", C.c_int32, [_P, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(GenSpec), C.POINTER(_P)]),
    ("qe_batch_nrows", C.c_int64, [_P]),
    ("qe_batch_ncols", C.c_int32, [_P]),
    ("qe_batch_column_type", C.c_int32, [_P, C.c_int32]),
    ("qe_batch_column_to_host", C.c_int32, [_P, _P, C.c_int32, C.c_int64, C.c_int64, _P, _P]),
    ("qe_batch_free", None, [_P, _P]),
    ("qe_csv_parse", C.c_int32, [_P, C.c_char_p, C.c_size_t, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(_P)]),
    ("qe_csv_parse_file", C.c_int32, [_P, C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(_P)]),
    ("qe_csv_nrows", C.c_int64, [_P]),
    ("qe_csv_ncols", C.c_int32, [_P]),
    ("qe_csv_column", C.c_int32, [_P, C.c_int32, C.POINTER(ColDesc)]),
    ("qe_csv_pin", C.c_int32, [_P, _P, C.POINTER(_P)]),
    ("qe_csv_free", None, [_P, _P]),
    ("qe_expr_compile", C.c_int32, [_P, C.c_char_p, C.c_size_t, C.POINTER(_P)]),
    ("qe_expr_result_type", C.c_int32, [_P]),
    ("qe_expr_free", None, [_P, _P]),
    ("qe_filter_project", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32, C.POINTER(_P)]),
    ("qe_filter_project_prepare", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32]),
    ("qe_filter_aggregate", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32,
                                        C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int64)]),
    ("qe_filter_groupby", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32, C.POINTER(_P)]),
    ("qe_filter_groupby_prepare", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32]),
    ("qe_filter_aggregate_prepare", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int32]),
    ("qe_result_count", C.c_int64, [_P]),
    ("qe_result_ncols", C.c_int32, [_P]),
    ("qe_result_column", C.c_int32, [_P, C.c_int32, C.POINTER(ColView)]),
    ("qe_result_column_to_host", C.c_int32, [_P, _P, C.c_int32, _P, _P]),
    ("qe_result_free", None, [_P, _P]),
    ("qe_result_to_host", C.c_int32, [_P, _P, C.POINTER(_P)]),
    ("qe_host_result_wait", C.c_int32, [_P, _P]),
    ("qe_host_result_count", C.c_int64, [_P]),
    ("qe_host_result_ncols", C.c_int32, [_P]),
    ("qe_host_result_column", C.c_int32, [_P, C.c_int32, C.POINTER(ColView)]),
    ("qe_host_result_free", None, [_P, _P]),
    ("qe_result_concat", C.c_int32, [_P, C.POINTER(_P), C.c_int32, C.POINTER(_P)]),
    ("qe_result_order_by", C.c_int32, [_P, _P, C.c_int32, C.POINTER(_P)]),
    ("qe_comm_unique_id", C.c_int32, [_P, _P]),
    ("qe_comm_init", C.c_int32, [_P, C.c_int32, C.c_int32, _P]),
    ("qe_comm_rank", C.c_int32, [_P]),
    ("qe_comm_nranks", C.c_int32, [_P]),
    ("qe_comm_destroy", None, [_P]),
    ("qe_gather", C.c_int32, [_P, _P, C.c_int32, C.POINTER(_P)]),
    ("qe_filter_project_gather", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("qe_comm_allgather_host", C.c_int32, [_P, _P, C.c_size_t, _P]),
    ("qe_filter_project_source", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32, C.POINTER(C.c_char_p)]),
    ("qe_filter_project_geometry", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("qe_filter_project_conjunct_order", C.c_int32, [_P, _P, _P, C.POINTER(_P), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    ("qe_stream_read_bandwidth", C.c_int32, [_P, C.c_int64, C.c_int32, C.POINTER(C.c_double)]),
    ("qe_stream_read_write_time", C.c_int32, [_P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
]

_lib: Optional[C.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile libqe_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", CSRC, "-j8"] + ([] if verbose else ["-s"]), check=True)
    return LIB_PATH


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  torch's libc10_hip.so asks for the unversioned name ``libamdhip64.so`` and finds the
    copy bundled in torch/lib; that request never matches a ``libamdhip64.so.7`` from /opt/rocm that libqe_hip.so pulled
    in earlier, so a process that creates a qe_ctx BEFORE importing torch (distributed.py needs torch) would end up with
    two runtimes, and the second one sees no GPU ("No HIP GPUs are available").  The other order works because torch's
    copy carries the SONAME libamdhip64.so.7 that libqe_hip.so asks for.  So: if torch is installed, load its bundled
    runtime first (without importing torch) and let libqe_hip.so bind to it."""
    if os.environ.get("QE_SYSTEM_HIP_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamd_comgr.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if not os.path.exists(path):
            return
    # the JIT compiler stays the image's own: hiprtc dlopens "libamd_comgr.so.3" by name and takes the first loaded
    # library with that SONAME, so the ROCm one goes in before torch's (older) copy
    system_comgr = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamd_comgr.so.3")
    try:
        if os.path.exists(system_comgr):
            C.CDLL(system_comgr, mode=C.RTLD_LOCAL)    # local: no symbol interposition into torch's runtime
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            C.CDLL(os.path.join(libdir, name), mode=C.RTLD_GLOBAL)
    except OSError:
        return


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(queryengine_amd has no CPU fallback)")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.qe_abi_version() != 1:
            raise RuntimeError("libqe_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(ctx, st: int) -> None:
    if st != OK:
        msg = lib().qe_last_error(ctx)
        raise QeError(st, msg.decode("utf-8", "replace") if msg else "")
