"""ctypes binding of libpls_hip.so (the C-ABI in include/pls_hip.h).

There is no CPU fallback: if the HIP extension has not been built, importing this module's
`lib()` raises, and every compute entry point of the library itself fails with
PLS_HIP_ERR_DEVICE when no gfx950 device is present.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLS_AMD_LIBRARY: another build of the same library -- the test suite's csrc/testing/libpls_hip.so, which carries the
# fault-injection hook of the exchange that the production library does not have
LIB_PATH = os.environ.get("PLS_AMD_LIBRARY") or os.path.join(_HERE, "csrc", "libpls_hip.so")

# enums of include/pls_hip.h
OK, ERR_INVALID, ERR_DEVICE, ERR_ALLOC, ERR_UNSUPPORTED, ERR_REDUCER = range(6)
KERNEL_TYPE1, KERNEL_TYPE2 = 0, 1          # PLS::METHOD, reference include/PLS/pls.h:131
F64, F32 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
ALGO_KERNEL, ALGO_NIPALS, ALGO_GRAM, ALGO_AUTO = 0, 1, 2, 3
OPT_ALGO, OPT_FUSE, OPT_PROFILE, OPT_POWER_ITERS, OPT_FUSED_GRID, OPT_WORK_LAYOUT, OPT_DEFER, OPT_GRAPH = 1, 2, 3, 4, 5, 6, 7, 8
REDUCE_SLICES = 8
XCHG_HANDLE_BYTES = 160  # PLS_HIP_XCHG_HANDLE_BYTES
FAM_XTY, FAM_XB, FAM_DEFLATE, FAM_FUSED, FAM_SMALL, FAM_COUNT = 0, 1, 2, 3, 4, 5
FAM_NAMES = ("xty", "xb", "deflate", "fused", "small")

_STATUS = {0: "OK", 1: "INVALID", 2: "DEVICE", 3: "ALLOC", 4: "UNSUPPORTED", 5: "REDUCER"}

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_int = ctypes.c_int

ALLREDUCE_FN = ctypes.CFUNCTYPE(_int, _vp, _vp, _i64, _vp)


class Timing(ctypes.Structure):
    _fields_ = [("fit_ms", ctypes.c_double),
                ("fits", _i64),
                ("fam_ms", ctypes.c_double * FAM_COUNT),
                ("fam_launches", _i64 * FAM_COUNT),
                ("fam_bytes", _i64 * FAM_COUNT)]


class PlsHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pls_hip status {code} ({_STATUS.get(code, '?')}): {msg}")
        self.code = code


# every symbol include/pls_hip.h declares: (name, restype, argtypes)
PROTOTYPES = [
    ("pls_hip_abi_version", _int, []),
    ("pls_hip_create", _int, [ctypes.POINTER(_vp), _int, _vp]),
    ("pls_hip_destroy", _int, [_vp]),
    ("pls_hip_set_stream", _int, [_vp, _vp]),
    ("pls_hip_set_option", _int, [_vp, _int, _i64]),
    ("pls_hip_get_option", _int, [_vp, _int, ctypes.POINTER(_i64)]),
    ("pls_hip_set_reducer", _int, [_vp, ALLREDUCE_FN, _vp, _int, _int]),
    ("pls_hip_set_reduce_buffer", _int, [_vp, _vp, _i64]),
    ("pls_hip_synchronize", _int, [_vp]),
    ("pls_hip_last_error", ctypes.c_char_p, [_vp]),
    ("pls_hip_get_timing", _int, [_vp, ctypes.POINTER(Timing)]),
    ("pls_hip_fit", _int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int, _int,
                           _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    ("pls_hip_coefficients", _int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp]),
    ("pls_hip_xb", _int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _int, _int, _vp, _i64]),
    ("pls_hip_xty", _int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _vp]),
    ("pls_hip_deflate", _int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _int]),
    ("pls_hip_colwise_z_scores", _int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp, _vp]),
    ("pls_hip_sse_by_components", _int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _int, _vp]),
    ("pls_hip_model_sse", _int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _int, _vp]),
    ("pls_hip_cv_folds", _int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _int, _int, _vp]),
    ("pls_hip_synth_x", _int, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_uint64, _int]),
    ("pls_hip_synth_y", _int, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_uint64, _int]),
    # one process, several GPUs: groups and resident matrices
    ("pls_hip_group_create", _int, [ctypes.POINTER(_vp), _int, ctypes.POINTER(_int)]),
    ("pls_hip_group_destroy", _int, [_vp]),
    ("pls_hip_xchg_create", _int, [_vp, _int, _int, _vp]),
    ("pls_hip_xchg_connect", _int, [_vp, _vp]),
    ("pls_hip_xchg_selftest", _int, [_vp]),
    ("pls_hip_xchg_destroy", _int, [_vp]),
    ("pls_hip_group_size", _int, [_vp]),
    ("pls_hip_group_exchange", _int, [_vp]),
    ("pls_hip_group_handle", _int, [_vp, _int, ctypes.POINTER(_vp)]),
    ("pls_hip_group_set_option", _int, [_vp, _int, _i64]),
    ("pls_hip_group_last_error", ctypes.c_char_p, [_vp]),
    ("pls_hip_group_upload", _int, [_vp, _vp, _i64, _i64, _i64, _int, ctypes.POINTER(_vp)]),
    ("pls_hip_group_upload_xy", _int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, ctypes.POINTER(_vp),
                                       ctypes.POINTER(_vp)]),
    ("pls_hip_group_alloc", _int, [_vp, _i64, _i64, _int, ctypes.POINTER(_vp)]),
    ("pls_hip_group_download", _int, [_vp, _vp, _i64, _i64, _vp, _i64]),
    ("pls_hip_group_free", _int, [_vp, _vp]),
    ("pls_hip_matrix_shape", _int, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_int)]),
    ("pls_hip_matrix_block", _int, [_vp, _int, ctypes.POINTER(_vp), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                    ctypes.POINTER(_i64)]),
    ("pls_hip_group_fit", _int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("pls_hip_group_xb", _int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    ("pls_hip_group_model_sse", _int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    ("pls_hip_group_cv_folds", _int, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp]),
]

_LIB = None


def lib() -> ctypes.CDLL:
    """Load libpls_hip.so (once).  Raises ImportError if the extension is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C pls_amd/csrc). "
            "pls_amd has no CPU fallback.")
    try:
        # torch ships its own libamdhip64.so.7; when torch is in the process it must be loaded
        # first so that this library binds to the SAME HIP runtime (device pointers and streams
        # are only meaningful inside one runtime instance).
        import torch  # noqa: F401
    except Exception:  # torch-free hosts (the C++ CLI, plain ctypes users) use /opt/rocm's runtime
        pass
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, res, args in PROTOTYPES:
        fn = getattr(L, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    if L.pls_hip_abi_version() != 1:
        raise ImportError("libpls_hip.so ABI version mismatch")
    _LIB = L
    return L


def last_error(handle) -> str:
    raw = lib().pls_hip_last_error(handle)
    return raw.decode("utf-8", "replace") if raw else ""


def check(rc: int, handle=None) -> None:
    if rc != OK:
        msg = ""
        if handle:
            raw = lib().pls_hip_last_error(handle)
            msg = raw.decode("utf-8", "replace") if raw else ""
        raise PlsHipError(rc, msg)
