"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes binding of the CPU oracle (oracle/qp_oracle.c): a plain-C restatement of
the reference's ``qp_solver::QPSolver<Scalar>`` (/root/reference/src/qp.cpp:11-386,
/root/reference/include/solvers/qp.hpp:118-248).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``sqp_solver_amd``) never
imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libqp_oracle.so")

SOLVED, MAX_ITER_EXCEEDED, UNSOLVED, NUMERICAL_ISSUES, UNINITIALIZED = range(5)
INEQUALITY_CONSTRAINT, EQUALITY_CONSTRAINT, LOOSE_BOUNDS = range(3)


class Settings(ctypes.Structure):
    """QPSolverSettings, qp.hpp:36-54."""

    _fields_ = [
        ("rho", ctypes.c_double),
        ("sigma", ctypes.c_double),
        ("alpha", ctypes.c_double),
        ("eps_rel", ctypes.c_double),
        ("eps_abs", ctypes.c_double),
        ("max_iter", ctypes.c_int),
        ("check_termination", ctypes.c_int),
        ("warm_start", ctypes.c_int),
        ("adaptive_rho", ctypes.c_int),
        ("adaptive_rho_tolerance", ctypes.c_double),
        ("adaptive_rho_interval", ctypes.c_int),
        ("verbose", ctypes.c_int),
    ]


class Info(ctypes.Structure):
    """QPSolverInfo, qp.hpp:72-80."""

    _fields_ = [
        ("status", ctypes.c_int),
        ("iter", ctypes.c_int),
        ("rho_updates", ctypes.c_int),
        ("_pad", ctypes.c_int),
        ("rho_estimate", ctypes.c_double),
        ("res_prim", ctypes.c_double),
        ("res_dual", ctypes.c_double),
    ]


INFO_DTYPE = np.dtype(
    [
        ("status", np.int32),
        ("iter", np.int32),
        ("rho_updates", np.int32),
        ("_pad", np.int32),
        ("rho_estimate", np.float64),
        ("res_prim", np.float64),
        ("res_dual", np.float64),
    ]
)
assert INFO_DTYPE.itemsize == ctypes.sizeof(Info) == 40

_lib = None


def build(force=False):
    """Compile oracle/libqp_oracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("qp_oracle.c", "qp_oracle_impl.h", "qp_oracle.h", "sqp_oracle.c", "sqp_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libqp_oracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


def _bind(_lib):
    _lib.qpo_default_settings.argtypes = [ctypes.POINTER(Settings)]
    _lib.qpo_max_threads.restype = ctypes.c_int
    for sfx, ct in (("_f64", ctypes.c_double), ("_f32", ctypes.c_float), ("_f80", ctypes.c_longdouble)):
        p = ctypes.POINTER(ct)
        pi = ctypes.POINTER(ctypes.c_int)
        g = lambda name: getattr(_lib, name + sfx)  # noqa: E731
        g("qpo_create").restype = ctypes.c_void_p
        g("qpo_destroy").argtypes = [ctypes.c_void_p]
        g("qpo_settings_ptr").restype = ctypes.POINTER(Settings)
        g("qpo_settings_ptr").argtypes = [ctypes.c_void_p]
        g("qpo_info_ptr").restype = ctypes.POINTER(Info)
        g("qpo_info_ptr").argtypes = [ctypes.c_void_p]
        g("qpo_set_legacy_cold_start").argtypes = [ctypes.c_void_p, ctypes.c_int]
        for name in ("qpo_primal", "qpo_dual", "qpo_z", "qpo_rho_vec"):
            g(name).restype = p
            g(name).argtypes = [ctypes.c_void_p]
        g("qpo_constr_type").restype = pi
        g("qpo_constr_type").argtypes = [ctypes.c_void_p]
        g("qpo_set_state").argtypes = [ctypes.c_void_p, p, p, p]
        g("qpo_constr_type_init").argtypes = [ctypes.c_int, p, p, pi]
        g("qpo_setup").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, p, p, p, p, p]
        g("qpo_update_qp").argtypes = [ctypes.c_void_p, p, p, p, p, p]
        g("qpo_solve").argtypes = [ctypes.c_void_p, p, p, p, p, p]
        g("qpo_solve_batch").argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, p, p, p, p, p,
            ctypes.POINTER(Settings), p, p, p, ctypes.POINTER(Info), ctypes.c_int,
        ]
        g("qpo_ldlt_factor_solve").restype = ctypes.c_int
        g("qpo_ldlt_factor_solve").argtypes = [ctypes.c_int, p, p, pi, p]
    return _lib


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(ctypes.CDLL(_LIB_PATH))
    return _lib


_native = None


def native_lib():
    """The timing copy built with -O3 -march=native ON THIS MACHINE (oracle/Makefile: libqp_oracle_native.so), or None when it cannot
    be built here.  Only bench.py's cpu_baseline leg asks for it; same sources, same results (no -ffast-math, no contraction)."""
    global _native
    if _native is None:
        path = os.path.join(_HERE, "libqp_oracle_native.so")
        try:
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libqp_oracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _native = _bind(ctypes.CDLL(path))
        except Exception:  # noqa: BLE001
            _native = False
    return _native or None


def default_settings(**kw):
    s = Settings()
    lib().qpo_default_settings(ctypes.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "_f64", ctypes.c_double
    if dtype == np.float32:
        return "_f32", ctypes.c_float
    if dtype == np.longdouble:
        return "_f80", ctypes.c_longdouble  # x87 extended precision yard-stick (not a reference instantiation)
    raise TypeError(dtype)


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _colmajor(M, dtype):
    """Eigen's default storage is column-major (qp.hpp:27)."""
    return np.asfortranarray(np.asarray(M, dtype=dtype))


class QPSolver:
    """Mirror of qp_solver::QPSolver<Scalar>'s public API (qp.hpp:148-173)."""

    def __init__(self, dtype=np.float64, legacy=False):
        self.dtype = np.dtype(dtype)
        self._sfx, self._ct = _sfx(dtype)
        self._L = lib()
        self._h = self._f("qpo_create")()
        self._f("qpo_set_legacy_cold_start")(self._h, 1 if legacy else 0)
        self.n = self.m = 0
        self._keep = None

    def _f(self, name):
        return getattr(self._L, name + self._sfx)

    def __del__(self):
        try:
            if self._h:
                self._f("qpo_destroy")(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def settings(self):
        return self._f("qpo_settings_ptr")(self._h).contents

    @property
    def info(self):
        return self._f("qpo_info_ptr")(self._h).contents

    def _prep(self, P, q, A, l, u):
        P = _colmajor(P, self.dtype)
        A = _colmajor(np.asarray(A, dtype=self.dtype).reshape(-1, P.shape[0]), self.dtype)
        q = np.ascontiguousarray(q, dtype=self.dtype)
        l = np.ascontiguousarray(l, dtype=self.dtype)
        u = np.ascontiguousarray(u, dtype=self.dtype)
        self._keep = (P, q, A, l, u)
        return tuple(_ptr(a, self._ct) for a in self._keep)

    def setup(self, P, q, A, l, u):
        P = np.asarray(P)
        A = np.asarray(A)
        self.n, self.m = P.shape[0], A.shape[0]
        ptrs = self._prep(P, q, A, l, u)
        self._f("qpo_setup")(self._h, self.n, self.m, *ptrs)

    def update_qp(self, P, q, A, l, u):
        self._f("qpo_update_qp")(self._h, *self._prep(P, q, A, l, u))

    def solve(self, P, q, A, l, u):
        self._f("qpo_solve")(self._h, *self._prep(P, q, A, l, u))

    def _vec(self, name, k):
        p = self._f(name)(self._h)
        return np.ctypeslib.as_array(p, shape=(k,)).copy() if k else np.zeros(0, self.dtype)

    def primal_solution(self):
        return self._vec("qpo_primal", self.n)

    def dual_solution(self):
        return self._vec("qpo_dual", self.m)

    def z(self):
        return self._vec("qpo_z", self.m)

    def rho_vec(self):
        return self._vec("qpo_rho_vec", self.m)

    def constr_type(self):
        p = self._f("qpo_constr_type")(self._h)
        return np.ctypeslib.as_array(p, shape=(self.m,)).copy()

    def set_state(self, x=None, z=None, y=None):
        def cv(a):
            return None if a is None else _ptr(np.ascontiguousarray(a, dtype=self.dtype), self._ct)

        self._f("qpo_set_state")(self._h, cv(x), cv(z), cv(y))


def constr_type_init(l, u, dtype=np.float64):
    """static QPSolver::constr_type_init, qp.cpp:283-294."""
    sfx, ct = _sfx(dtype)
    l = np.ascontiguousarray(l, dtype=dtype)
    u = np.ascontiguousarray(u, dtype=dtype)
    out = np.zeros(l.shape[0], dtype=np.int32)
    getattr(lib(), "qpo_constr_type_init" + sfx)(
        l.shape[0], _ptr(l, ct), _ptr(u, ct), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    )
    return out


def solve_batch(P, q, A, l, u, settings=None, nthreads=0, dtype=np.float64, native=False):
    """setup()+solve() for every QP of a batch (cold start, fresh solver each).

    Layout: P[b] (n,n), A[b] (m,n) as numpy arrays indexed [b, i, j]; they are
    converted to per-QP column-major storage (the reference's Eigen layout).
    Returns x (B,n), y (B,m), z (B,m), info (structured array).
    """
    sfx, ct = _sfx(dtype)
    P = np.asarray(P, dtype=dtype)
    A = np.asarray(A, dtype=dtype)
    B, n = P.shape[0], P.shape[1]
    m = A.shape[1]
    Pc = np.ascontiguousarray(np.transpose(P, (0, 2, 1)))  # [b][j][i] == col-major per QP
    Ac = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))
    q = np.ascontiguousarray(q, dtype=dtype)
    l = np.ascontiguousarray(l, dtype=dtype)
    u = np.ascontiguousarray(u, dtype=dtype)
    x = np.zeros((B, n), dtype=dtype)
    y = np.zeros((B, m), dtype=dtype)
    z = np.zeros((B, m), dtype=dtype)
    info = np.zeros(B, dtype=INFO_DTYPE)
    if settings is None:
        settings = default_settings()
    L = (native_lib() if native else None) or lib()  # native: the -O3 -march=native timing copy where it can be built
    getattr(L, "qpo_solve_batch" + sfx)(
        n, m, B, _ptr(Pc, ct), _ptr(q, ct), _ptr(Ac, ct), _ptr(l, ct), _ptr(u, ct),
        ctypes.byref(settings), _ptr(x, ct), _ptr(y, ct), _ptr(z, ct),
        info.ctypes.data_as(ctypes.POINTER(Info)), int(nthreads),
    )
    return x, y, z, info


def ldlt_factor_solve(K, rhs=None, dtype=np.float64):
    """Factor symmetric K (lower triangle read) with the oracle's pivoted LDL^T."""
    sfx, ct = _sfx(dtype)
    K = np.asfortranarray(np.asarray(K, dtype=dtype))
    N = K.shape[0]
    L = np.zeros((N, N), dtype=dtype, order="F")
    tr = np.zeros(N, dtype=np.int32)
    r = None if rhs is None else np.ascontiguousarray(rhs, dtype=dtype).copy()
    ok = getattr(lib(), "qpo_ldlt_factor_solve" + sfx)(
        N, _ptr(K, ct), _ptr(L, ct), tr.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
        None if r is None else _ptr(r, ct),
    )
    return bool(ok), L, tr, r


def max_threads():
    return lib().qpo_max_threads()
