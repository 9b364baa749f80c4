"""ctypes view of the C-ABI in include/sqp_hip.h (libsqp_hip.so)."""
import ctypes
import os

import numpy as np

from . import build as _build

F64, F32 = 0, 1
HOST, DEVICE = 0, 1
SOLVED, MAX_ITER_EXCEEDED, UNSOLVED, NUMERICAL_ISSUES, UNINITIALIZED = range(5)
INEQUALITY_CONSTRAINT, EQUALITY_CONSTRAINT, LOOSE_BOUNDS = range(3)
FLAG_LEGACY_COLD_START, FLAG_FORCE_GENERIC, FLAG_CSR_EXPAND, FLAG_KEEP_FACTOR, FLAG_F32_ARITH = 1, 2, 8, 16, 32
OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE = 0, -1, -2, -3, -4

# every symbol include/sqp_hip.h declares
SYMBOLS = [
    "sqph_default_settings", "sqph_create", "sqph_destroy", "sqph_set_stream", "sqph_set_settings",
    "sqph_get_settings", "sqph_setup", "sqph_update_qp", "sqph_solve", "sqph_setup_solve",
    "sqph_get_solution", "sqph_set_state", "sqph_device_state", "sqph_synchronize", "sqph_kernel_name",
    "sqph_enable_timing", "sqph_last_kernel_ms", "sqph_collect_kernel_ms", "sqph_last_error", "sqph_global_error",
    "sqph_constr_type_init", "sqph_algorithmic_bytes", "sqph_version",
    "sqph_setup_csr", "sqph_update_qp_csr", "sqph_solve_csr", "sqph_setup_solve_csr",
    "sqph_shard_bounds", "sqph_device_count", "sqph_own_stream", "sqph_gather_create", "sqph_gather_create_ex", "sqph_gather_transport", "sqph_gather_destroy",
    "sqph_gather_post", "sqph_gather_fetch", "sqph_gather_device_ptrs", "sqph_setup_solve_reuse", "sqph_set_trace_qp", "sqph_get_trace",
    "sqph_update_solve", "sqph_gather_post_many",
    "sqph_setup_csr_sp", "sqph_update_qp_csr_sp", "sqph_solve_csr_sp", "sqph_setup_solve_csr_sp",
    "sqph_update_solve_csr", "sqph_update_solve_csr_sp", "sqph_setup_solve_reuse_csr", "sqph_setup_solve_reuse_csr_sp",
]


class Settings(ctypes.Structure):
    """sqph_settings == QPSolverSettings (reference include/solvers/qp.hpp:36-54)."""

    _fields_ = [
        ("rho", ctypes.c_double), ("sigma", ctypes.c_double), ("alpha", ctypes.c_double),
        ("eps_rel", ctypes.c_double), ("eps_abs", ctypes.c_double),
        ("max_iter", ctypes.c_int), ("check_termination", ctypes.c_int),
        ("warm_start", ctypes.c_int), ("adaptive_rho", ctypes.c_int),
        ("adaptive_rho_tolerance", ctypes.c_double),
        ("adaptive_rho_interval", ctypes.c_int), ("verbose", ctypes.c_int),
    ]


class Info(ctypes.Structure):
    """sqph_info == QPSolverInfo (reference include/solvers/qp.hpp:72-80)."""

    _fields_ = [
        ("status", ctypes.c_int), ("iter", ctypes.c_int), ("rho_updates", ctypes.c_int), ("_pad", ctypes.c_int),
        ("rho_estimate", ctypes.c_double), ("res_prim", ctypes.c_double), ("res_dual", ctypes.c_double),
    ]


INFO_DTYPE = np.dtype([
    ("status", np.int32), ("iter", np.int32), ("rho_updates", np.int32), ("_pad", np.int32),
    ("rho_estimate", np.float64), ("res_prim", np.float64), ("res_dual", np.float64),
])
assert INFO_DTYPE.itemsize == ctypes.sizeof(Info) == 40


class QPBatch(ctypes.Structure):
    """sqph_qp_batch == a batch of QuadraticProblem (reference include/solvers/qp.hpp:19-34)."""

    _fields_ = [
        ("batch", ctypes.c_int), ("memspace", ctypes.c_int),
        ("P", ctypes.c_void_p), ("q", ctypes.c_void_p), ("A", ctypes.c_void_p),
        ("l", ctypes.c_void_p), ("u", ctypes.c_void_p),
        ("stride_P", ctypes.c_longlong), ("stride_q", ctypes.c_longlong), ("stride_A", ctypes.c_longlong),
        ("stride_l", ctypes.c_longlong), ("stride_u", ctypes.c_longlong),
    ]


class CsrBatch(ctypes.Structure):
    """sqph_csr_batch: the same batch with A in CSR (reference include/unsupported/qp_solver.hpp:17-32)."""

    _fields_ = [
        ("batch", ctypes.c_int), ("memspace", ctypes.c_int),
        ("P", ctypes.c_void_p), ("q", ctypes.c_void_p),
        ("A_rowptr", ctypes.c_void_p), ("A_colind", ctypes.c_void_p), ("A_val", ctypes.c_void_p),
        ("l", ctypes.c_void_p), ("u", ctypes.c_void_p),
        ("stride_P", ctypes.c_longlong), ("stride_q", ctypes.c_longlong), ("stride_rowptr", ctypes.c_longlong),
        ("stride_colind", ctypes.c_longlong), ("stride_val", ctypes.c_longlong),
        ("stride_l", ctypes.c_longlong), ("stride_u", ctypes.c_longlong), ("nnz_max", ctypes.c_longlong),
    ]


class CscP(ctypes.Structure):
    """sqph_csc_P: P of a CsrBatch in compressed-column form (the Eigen::SparseMatrix P of the legacy sparse class,
    reference include/unsupported/qp_solver.hpp:24-25)."""

    _fields_ = [
        ("colptr", ctypes.c_void_p), ("rowind", ctypes.c_void_p), ("val", ctypes.c_void_p),
        ("stride_colptr", ctypes.c_longlong), ("stride_rowind", ctypes.c_longlong), ("stride_val", ctypes.c_longlong),
        ("nnz_max", ctypes.c_longlong),
    ]


_lib = None


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load libsqp_hip.so. Raises if it cannot be built/loaded: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and _build.needs_build():
        _build.build()
    if not os.path.exists(_build.LIB):
        raise RuntimeError("libsqp_hip.so is missing (%s); run `python -m sqp_solver_amd.build`" % _build.LIB)
    L = ctypes.CDLL(_build.LIB)
    vp, i, pi = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)
    L.sqph_default_settings.argtypes = [ctypes.POINTER(Settings)]
    L.sqph_default_settings.restype = None
    L.sqph_create.argtypes = [ctypes.POINTER(vp), i, i, i, i, i, i]
    L.sqph_destroy.argtypes = [vp]
    L.sqph_destroy.restype = None
    L.sqph_set_stream.argtypes = [vp, vp]
    L.sqph_set_settings.argtypes = [vp, ctypes.POINTER(Settings)]
    L.sqph_get_settings.argtypes = [vp, ctypes.POINTER(Settings)]
    for name in ("sqph_setup", "sqph_update_qp", "sqph_solve", "sqph_setup_solve", "sqph_update_solve"):
        getattr(L, name).argtypes = [vp, ctypes.POINTER(QPBatch)]
    for name in ("sqph_setup_csr", "sqph_update_qp_csr", "sqph_solve_csr", "sqph_setup_solve_csr", "sqph_update_solve_csr", "sqph_setup_solve_reuse_csr"):
        getattr(L, name).argtypes = [vp, ctypes.POINTER(CsrBatch)]
        getattr(L, name + "_sp").argtypes = [vp, ctypes.POINTER(CsrBatch), ctypes.POINTER(CscP)]
    L.sqph_get_solution.argtypes = [vp, i, i, vp, vp, vp, vp]
    L.sqph_set_state.argtypes = [vp, i, i, vp, vp, vp]
    L.sqph_device_state.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.sqph_synchronize.argtypes = [vp]
    L.sqph_kernel_name.argtypes = [vp]
    L.sqph_kernel_name.restype = ctypes.c_char_p
    L.sqph_enable_timing.argtypes = [vp, i]
    L.sqph_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.sqph_collect_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), i, pi]
    L.sqph_last_error.argtypes = [vp]
    L.sqph_last_error.restype = ctypes.c_char_p
    L.sqph_global_error.restype = ctypes.c_char_p
    L.sqph_constr_type_init.argtypes = [i, i, vp, vp, pi]
    L.sqph_algorithmic_bytes.argtypes = [i, i, i]
    L.sqph_algorithmic_bytes.restype = ctypes.c_longlong
    L.sqph_version.restype = i
    _lib = L
    return L
