"""TEST INFRASTRUCTURE: drive the product's HIP kernels under the host SIMT emulator (tests/sim/)."""
import ctypes
import os
import subprocess

import numpy as np

from sqp_solver_amd import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
SIMDIR = os.path.join(HERE, "sim")
LIB = os.path.join(SIMDIR, "libsqph_sim.so")

MODE_SETUP, MODE_UPDATE, MODE_SOLVE, MODE_COLD_RESET, MODE_NO_FACTOR_STORE, MODE_REFACTOR, MODE_SAME_MATRICES = 1, 2, 4, 8, 16, 32, 64
GENERIC, WG, CSR, G16, LANE, LANE_F32 = 0, 2, 3, 4, 6, 7
GENERIC_F32_ARITH = 9  # measurement only: the generic kernel in fp32 arithmetic (fp32 state arrays)
WG_STACK = 12  # register-tiled kernel on the stacked operator (m <= 104 at the C3 shape)
CSR_DENSE = 11  # the sparse kernel's dense-A mode (A streamed from global memory, W in the CU's registers)
LANE_QUAD = 13  # the one-QP-per-lane kernel's quad variant (four lanes per QP, m <= 4: small batches)
CSRB = 14  # the block-row sparse kernel (admm_csrb_kernel.h): 512 lanes per QP, W as MFMA blocks in registers
WG_F32 = 10  # register-tiled kernels with fp32 products (SQPH_FLAG_F32_ARITH), float interface only


class SimArgs(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int), ("m", ctypes.c_int), ("batch", ctypes.c_int), ("mode", ctypes.c_int),
        ("P", ctypes.c_void_p), ("q", ctypes.c_void_p), ("A", ctypes.c_void_p), ("l", ctypes.c_void_p), ("u", ctypes.c_void_p),
        ("sP", ctypes.c_longlong), ("sq", ctypes.c_longlong), ("sA", ctypes.c_longlong), ("sl", ctypes.c_longlong), ("su", ctypes.c_longlong),
        ("x", ctypes.c_void_p), ("z", ctypes.c_void_p), ("y", ctypes.c_void_p), ("rho_vec", ctypes.c_void_p),
        ("ctype", ctypes.c_void_p), ("rho", ctypes.c_void_p), ("info", ctypes.c_void_p),
        ("Sinv", ctypes.c_void_p), ("At", ctypes.c_void_p),
        ("rho0", ctypes.c_double), ("sigma", ctypes.c_double), ("alpha", ctypes.c_double),
        ("eps_rel", ctypes.c_double), ("eps_abs", ctypes.c_double), ("rho_tol", ctypes.c_double),
        ("max_iter", ctypes.c_int), ("check_termination", ctypes.c_int), ("warm_start", ctypes.c_int),
        ("adaptive_rho", ctypes.c_int), ("adaptive_rho_interval", ctypes.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", SIMDIR, "-s"])
        _lib = ctypes.CDLL(LIB)
        _lib.sim_run.argtypes = [ctypes.POINTER(SimArgs), ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.sim_run.restype = ctypes.c_int
        _lib.sim_run_csr.argtypes = [ctypes.POINTER(SimArgs), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                     ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
        _lib.sim_run_csr.restype = ctypes.c_int
        _lib.sim_run_csrb.argtypes = _lib.sim_run_csr.argtypes
        _lib.sim_run_csrb.restype = ctypes.c_int
        _lib.sim_run_csrb_sp.argtypes = _lib.sim_run_csr.argtypes + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                                                     ctypes.c_longlong, ctypes.c_longlong]
        _lib.sim_run_csrb_sp.restype = ctypes.c_int
    return _lib


class SimSolverBatch:
    """Same surface as sqp_solver_amd.QPSolverBatch, executed by the emulator (host arrays)."""

    def __init__(self, n, m, batch, dtype=np.float64, variant=GENERIC, nt=64, legacy_cold_start=False, keep_factor=False):
        self.n, self.m, self.batch = n, m, batch
        self.dtype = np.dtype(dtype)
        self.variant, self.nt = variant, nt
        self.legacy = legacy_cold_start
        # factor residency policy of the host library (capi.hip launch_typed), mirrored
        self.keep_factor, self.factor_resident = keep_factor, False
        self.settings = _capi.Settings()
        s = self.settings
        s.rho, s.sigma, s.alpha, s.eps_rel, s.eps_abs = 0.1, 1e-6, 1.0, 1e-3, 1e-3
        s.max_iter, s.check_termination, s.warm_start, s.adaptive_rho = 1000, 25, 0, 0
        s.adaptive_rho_tolerance, s.adaptive_rho_interval, s.verbose = 5, 25, 0
        mm = max(m, 1)
        # solver state is fp64 whatever the interface Scalar is (QPSolver<float>)
        sdt = np.float32 if variant == GENERIC_F32_ARITH else np.float64
        self.x = np.zeros((batch, n), sdt)
        self.zv = np.zeros((batch, mm), sdt)
        self.y = np.zeros((batch, mm), sdt)
        self.rho_vec = np.zeros((batch, mm), sdt)
        self.ctype = np.zeros((batch, mm), np.int32)
        self.rho = np.zeros(batch, sdt)
        self.info_arr = np.zeros(batch, _capi.INFO_DTYPE)
        self.info_arr["status"] = 4
        self.Sinv = np.zeros((batch, 2 * n * n), sdt)
        self.At = np.zeros((batch, mm * n), sdt)

    def _run(self, mode, P, q, A, l, u, csr=None):
        n, m = self.n, self.m
        sparse_P = None
        if isinstance(P, tuple):  # (colptr, rowind, val): the sparse-P instantiations of the block-row kernel
            assert csr is not None and self.variant == CSRB
            sparse_P = P
            P = np.zeros((np.asarray(q).shape[0], n, n)) if np.asarray(q).ndim == 2 else np.zeros((n, n))
        if csr is not None:
            A = np.zeros((np.asarray(P).shape[0], m, n)) if np.asarray(P).ndim == 3 else np.zeros((m, n))
        if m == 0:
            B0 = np.asarray(P).shape[0] if np.asarray(P).ndim == 3 else self.batch
            A = np.zeros((B0, 0, n)) if A is None else A
            l = np.zeros((B0, 0)) if l is None else l
            u = np.zeros((B0, 0)) if u is None else u
        P = np.ascontiguousarray(np.swapaxes(np.asarray(P, self.dtype), -1, -2))
        A = np.ascontiguousarray(np.swapaxes(np.asarray(A, self.dtype), -1, -2))
        q = np.ascontiguousarray(q, self.dtype)
        l = np.ascontiguousarray(l, self.dtype)
        u = np.ascontiguousarray(u, self.dtype)
        B = P.shape[0] if P.ndim == 3 else self.batch
        a = SimArgs()
        a.n, a.m, a.batch = n, m, B
        a.mode = mode | (MODE_COLD_RESET if self.legacy else 0)
        fused = bool(mode & (MODE_SETUP | MODE_UPDATE)) and bool(mode & MODE_SOLVE)
        if fused and not self.keep_factor:
            a.mode |= MODE_NO_FACTOR_STORE
        if not (mode & (MODE_SETUP | MODE_UPDATE)) and not self.factor_resident:
            a.mode |= MODE_REFACTOR
        if (mode & MODE_SAME_MATRICES) and not self.factor_resident:
            a.mode &= ~MODE_SAME_MATRICES
        if (mode & (MODE_SETUP | MODE_UPDATE)) or (a.mode & MODE_REFACTOR):
            self.factor_resident = self.variant in (GENERIC, GENERIC_F32_ARITH) or not (a.mode & MODE_NO_FACTOR_STORE)
        for name, arr, per in (("P", P, n * n), ("q", q, n), ("A", A, m * n), ("l", l, m), ("u", u, m)):
            setattr(a, name, arr.ctypes.data)
            shared = arr.ndim == (2 if name in ("P", "A") else 1)
            setattr(a, "s" + name, 0 if shared else per)
        a.x, a.z, a.y = self.x.ctypes.data, self.zv.ctypes.data, self.y.ctypes.data
        a.rho_vec, a.ctype, a.rho = self.rho_vec.ctypes.data, self.ctype.ctypes.data, self.rho.ctypes.data
        a.info, a.Sinv, a.At = self.info_arr.ctypes.data, self.Sinv.ctypes.data, self.At.ctypes.data
        s = self.settings
        a.rho0, a.sigma, a.alpha, a.eps_rel, a.eps_abs = s.rho, s.sigma, s.alpha, s.eps_rel, s.eps_abs
        a.rho_tol = s.adaptive_rho_tolerance
        a.max_iter, a.check_termination, a.warm_start = s.max_iter, s.check_termination, s.warm_start
        a.adaptive_rho, a.adaptive_rho_interval = s.adaptive_rho, s.adaptive_rho_interval
        if csr is not None:
            rp, ci, v = csr
            rp = np.ascontiguousarray(rp, np.int32)
            ci = np.ascontiguousarray(ci, np.int32)
            v = np.ascontiguousarray(v, self.dtype)
            shared = rp.ndim == 1
            fn = lib().sim_run_csrb if self.variant == CSRB else lib().sim_run_csr
            extra = ()
            if sparse_P is not None:
                pc = np.ascontiguousarray(sparse_P[0], np.int32)
                pr = np.ascontiguousarray(sparse_P[1], np.int32)
                pv = np.ascontiguousarray(sparse_P[2], self.dtype)
                fn = lib().sim_run_csrb_sp
                extra = (pc.ctypes.data, pr.ctypes.data, pv.ctypes.data, 0 if pc.ndim == 1 else pc.shape[-1],
                         0 if pr.ndim == 1 else pr.shape[-1], 0 if pv.ndim == 1 else pv.shape[-1])
            rc = fn(ctypes.byref(a), rp.ctypes.data, ci.ctypes.data, v.ctypes.data, 0 if shared else rp.shape[-1],
                                   0 if shared else ci.shape[-1], 0 if v.ndim == 1 else v.shape[-1], ci.shape[-1],
                                   1 if self.dtype == np.float32 else 0, *extra)
            if rc != 0:
                raise RuntimeError("sim_run_csr failed rc=%d" % rc)
            self._last = B
            return
        rc = lib().sim_run(ctypes.byref(a), self.variant, 1 if self.dtype == np.float32 else 0, self.nt)
        if rc != 0:
            raise RuntimeError("sim_run failed rc=%d" % rc)
        self._last = B

    def setup(self, P, q, A, l, u):
        self._run(MODE_SETUP, P, q, A, l, u)

    def update_qp(self, P, q, A, l, u):
        self._run(MODE_UPDATE, P, q, A, l, u)

    def solve(self, P, q, A, l, u):
        self._run(MODE_SOLVE, P, q, A, l, u)

    def setup_solve(self, P, q, A, l, u):
        self._run(MODE_SETUP | MODE_SOLVE, P, q, A, l, u)

    def update_solve(self, P, q, A, l, u):
        self._run(MODE_UPDATE | MODE_SOLVE, P, q, A, l, u)

    def setup_solve_reuse(self, P, q, A, l, u):
        self._run(MODE_SETUP | MODE_SOLVE | MODE_SAME_MATRICES, P, q, A, l, u)

    def setup_csr(self, P, q, rp, ci, v, l, u):
        self._run(MODE_SETUP, P, q, None, l, u, csr=(rp, ci, v))

    def update_qp_csr(self, P, q, rp, ci, v, l, u):
        self._run(MODE_UPDATE, P, q, None, l, u, csr=(rp, ci, v))

    def solve_csr(self, P, q, rp, ci, v, l, u):
        self._run(MODE_SOLVE, P, q, None, l, u, csr=(rp, ci, v))

    def update_solve_csr(self, P, q, rp, ci, v, l, u):
        self._run(MODE_UPDATE | MODE_SOLVE, P, q, None, l, u, csr=(rp, ci, v))

    def setup_solve_reuse_csr(self, P, q, rp, ci, v, l, u):
        self._run(MODE_SETUP | MODE_SOLVE | MODE_SAME_MATRICES, P, q, None, l, u, csr=(rp, ci, v))

    def setup_solve_csr(self, P, q, rp, ci, v, l, u):
        self._run(MODE_SETUP | MODE_SOLVE, P, q, None, l, u, csr=(rp, ci, v))

    def primal_solution(self):
        return self.x[: self._last].astype(self.dtype)

    def dual_solution(self):
        return self.y[: self._last, : self.m].astype(self.dtype)

    def z(self):
        return self.zv[: self._last, : self.m].astype(self.dtype)

    def info(self):
        return self.info_arr[: self._last].copy().view(np.recarray)

    def solution(self):
        return self.primal_solution(), self.dual_solution(), self.z(), self.info()

    def set_state(self, x=None, z=None, y=None):
        if x is not None:
            self.x[:] = x
        if z is not None:
            self.zv[:, : self.m] = z
        if y is not None:
            self.y[:, : self.m] = y
