"""Batched host-side mirror of the reference's ``qp_solver::QPSolver<Scalar>``.

One ``QPSolverBatch`` owns ``batch`` independent solver instances for same-(n, m) problems and
drives them through the C-ABI of libsqp_hip.so (include/sqp_hip.h).  Method names, argument
meaning and status/iteration bookkeeping follow the reference class
(/root/reference/include/solvers/qp.hpp:148-173, /root/reference/src/qp.cpp):

    setup(qp)      -> setup(P, q, A, l, u)
    update_qp(qp)  -> update_qp(P, q, A, l, u)
    solve(qp)      -> solve(P, q, A, l, u)
    primal_solution() / dual_solution() / settings / info()

``setup_solve`` is the fused single-launch form of ``setup(); solve()`` that the reference's
only production caller performs per subproblem (src/sqp.cpp:221-222).

Problem arrays may be numpy arrays (host; copied to the GPU inside the call) or torch CUDA
tensors (used in place, zero copy).  There is no CPU execution path: without a HIP device the
constructor raises.
"""
import ctypes

import numpy as np

from . import _capi

SOLVED = _capi.SOLVED
MAX_ITER_EXCEEDED = _capi.MAX_ITER_EXCEEDED
UNSOLVED = _capi.UNSOLVED
NUMERICAL_ISSUES = _capi.NUMERICAL_ISSUES
UNINITIALIZED = _capi.UNINITIALIZED
INEQUALITY_CONSTRAINT = _capi.INEQUALITY_CONSTRAINT
EQUALITY_CONSTRAINT = _capi.EQUALITY_CONSTRAINT
LOOSE_BOUNDS = _capi.LOOSE_BOUNDS

# public constants of the reference class, qp.hpp:136-141
RHO_MIN = 1e-6
RHO_MAX = 1e6
RHO_TOL = 1e-4
RHO_EQ_FACTOR = 1e3
LOOSE_BOUNDS_THRESH = 1e16


class SqphError(RuntimeError):
    pass


def default_settings(**kw):
    s = _capi.Settings()
    _capi.load().sqph_default_settings(ctypes.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def constr_type_init(l, u, dtype=np.float64):
    """static QPSolver::constr_type_init(l, u, constr_type) — src/qp.cpp:283-294."""
    dtype = np.dtype(dtype)
    l = np.ascontiguousarray(l, dtype=dtype)
    u = np.ascontiguousarray(u, dtype=dtype)
    out = np.zeros(l.shape[0], dtype=np.int32)
    rc = _capi.load().sqph_constr_type_init(
        _capi.F32 if dtype == np.float32 else _capi.F64, l.shape[0], l.ctypes.data, u.ctypes.data,
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if rc != 0:
        raise SqphError(_capi.load().sqph_global_error().decode())
    return out


class QPSolverBatch:
    def __init__(self, n, m, batch, dtype=np.float64, device=0, legacy_cold_start=False, force_generic=False, csr_expand=False,
                 keep_factor=False, f32_arith=False):
        self._L = _capi.load()
        self.n, self.m, self.batch = int(n), int(m), int(batch)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError("dtype must be float64 or float32")
        self._dt = _capi.F32 if self.dtype == np.float32 else _capi.F64
        self.device = int(device)
        flags = ((_capi.FLAG_LEGACY_COLD_START if legacy_cold_start else 0) | (_capi.FLAG_FORCE_GENERIC if force_generic else 0)
                 | (_capi.FLAG_CSR_EXPAND if csr_expand else 0) | (_capi.FLAG_KEEP_FACTOR if keep_factor else 0) | (_capi.FLAG_F32_ARITH if f32_arith else 0))
        h = ctypes.c_void_p()
        rc = self._L.sqph_create(ctypes.byref(h), self.device, self.n, self.m, self.batch, self._dt, flags)
        if rc != 0:
            raise SqphError("sqph_create failed (%d): %s" % (rc, self._L.sqph_global_error().decode()))
        self._h = h
        self.settings = default_settings()
        self._keep = None
        self._last_batch = 0

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._L.sqph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SqphError("%s failed (%d): %s" % (what, rc, self._L.sqph_last_error(self._h).decode()))

    def set_stream(self, stream_ptr):
        """Bind to a hipStream_t (integer handle, e.g. torch.cuda.current_stream().cuda_stream)."""
        self._check(self._L.sqph_set_stream(self._h, ctypes.c_void_p(stream_ptr)), "sqph_set_stream")

    def _prep_one(self, a, per_qp_shape, colmajor_in):
        """Returns (keepalive, pointer, stride_elems, is_device, batch or None)."""
        nd = len(per_qp_shape)
        if _is_torch(a):
            import torch

            tdt = torch.float32 if self.dtype == np.float32 else torch.float64
            if a.dtype != tdt:
                raise TypeError("tensor dtype %s does not match solver dtype %s" % (a.dtype, self.dtype))
            if not a.is_cuda:
                a = a.detach().cpu().numpy()
            else:
                shared = a.dim() == nd
                if nd == 2 and not colmajor_in:
                    a = a.transpose(-1, -2)
                a = a.contiguous()
                b = None if shared else a.shape[0]
                return a, a.data_ptr(), 0 if shared else int(np.prod(per_qp_shape)), True, b
        a = np.asarray(a, dtype=self.dtype)
        shared = a.ndim == nd
        if nd == 2 and not colmajor_in:
            a = np.swapaxes(a, -1, -2)
        a = np.ascontiguousarray(a)
        b = None if shared else a.shape[0]
        return a, a.ctypes.data, 0 if shared else int(np.prod(per_qp_shape)), False, b

    def _batch_desc(self, P, q, A, l, u, colmajor):
        n, m = self.n, self.m
        shapes = {"P": (n, n), "q": (n,), "A": (m, n), "l": (m,), "u": (m,)}
        items = {}
        dev = None
        batch = None
        for name, arr in (("P", P), ("q", q), ("A", A), ("l", l), ("u", u)):
            if arr is None:
                if m == 0 and name in ("A", "l", "u"):
                    items[name] = (None, None, 0, None, None)
                    continue
                raise ValueError("%s is required" % name)
            shp = shapes[name]
            if colmajor and name == "A":
                shp_in = (n, m)
            else:
                shp_in = shp
            if tuple(arr.shape[-len(shp):]) != tuple(shp_in):
                raise ValueError("%s has shape %s, expected [...,%s]" % (name, tuple(arr.shape), shp_in))
            k, ptr, stride, is_dev, b = self._prep_one(arr, shp, colmajor)
            items[name] = (k, ptr, stride, is_dev, b)
            if dev is None:
                dev = is_dev
            elif dev != is_dev:
                raise ValueError("mixing host and device problem arrays is not supported")
            if b is not None:
                if batch is None:
                    batch = b
                elif batch != b:
                    raise ValueError("inconsistent batch sizes")
        if batch is None:
            batch = self.batch
        if batch > self.batch:
            raise ValueError("batch %d exceeds capacity %d" % (batch, self.batch))
        d = _capi.QPBatch()
        d.batch = batch
        d.memspace = _capi.DEVICE if dev else _capi.HOST
        for name in ("P", "q", "A", "l", "u"):
            setattr(d, name, items[name][1])
            setattr(d, "stride_" + name, items[name][2])
        self._keep = [items[k][0] for k in items]
        if dev:
            import torch

            self.set_stream(torch.cuda.current_stream().cuda_stream)
        self._last_batch = batch
        return d

    def _push_settings(self):
        self._check(self._L.sqph_set_settings(self._h, ctypes.byref(self.settings)), "sqph_set_settings")

    def _call(self, fn, what, P, q, A, l, u, colmajor):
        self._push_settings()
        d = self._batch_desc(P, q, A, l, u, colmajor)
        self._check(fn(self._h, ctypes.byref(d)), what)

    # ------------------------------------------------------------------ reference API
    def setup(self, P, q, A, l, u, colmajor=False):
        """QPSolver::setup (src/qp.cpp:11-44) for every QP of the batch."""
        self._call(self._L.sqph_setup, "sqph_setup", P, q, A, l, u, colmajor)

    def update_qp(self, P, q, A, l, u, colmajor=False):
        """QPSolver::update_qp (src/qp.cpp:46-62)."""
        self._call(self._L.sqph_update_qp, "sqph_update_qp", P, q, A, l, u, colmajor)

    def solve(self, P, q, A, l, u, colmajor=False):
        """QPSolver::solve (src/qp.cpp:64-157)."""
        self._call(self._L.sqph_solve, "sqph_solve", P, q, A, l, u, colmajor)

    def setup_solve(self, P, q, A, l, u, colmajor=False):
        """setup(); solve() in one kernel launch (what SQP::run_solve_qp does, src/sqp.cpp:221-222)."""
        self._call(self._L.sqph_setup_solve, "sqph_setup_solve", P, q, A, l, u, colmajor)

    def update_solve(self, P, q, A, l, u, colmajor=False):
        """update_qp(); solve() in one launch: new matrices and bounds, the iterates of the previous call kept (src/qp.cpp:46-62)."""
        self._call(self._L.sqph_update_solve, "sqph_update_solve", P, q, A, l, u, colmajor)

    def set_trace_qp(self, index):
        self._check(self._L.sqph_set_trace_qp(self._h, int(index)), "sqph_set_trace_qp")

    def trace(self):
        """settings.verbose: the records of the last solve call for the traced QP, one row per termination check:
        [iter, objective 0.5 x'Px + q'x, res_prim, res_dual] (what the reference's print_status prints, src/qp.cpp:373-383)."""
        cnt = ctypes.c_int()
        self._check(self._L.sqph_get_trace(self._h, None, 0, ctypes.byref(cnt)), "sqph_get_trace")
        rec = np.zeros((cnt.value, 4))
        if cnt.value:
            self._check(self._L.sqph_get_trace(self._h, rec.ctypes.data_as(ctypes.c_void_p), cnt.value, ctypes.byref(cnt)), "sqph_get_trace")
        return rec

    def setup_solve_reuse(self, P, q, A, l, u, colmajor=False):
        """setup(); solve() for QPs whose P and A are those of the previous setup (only q, l, u differ): the factor is rebuilt
        only where the rho vector changed (the SQP driver's second-order correction, src/sqp.cpp:244-276)."""
        self._call(self._L.sqph_setup_solve_reuse, "sqph_setup_solve_reuse", P, q, A, l, u, colmajor)

    # ------------------------------------------------------------------ CSR-A variants (BASELINE config 5)
    def _csr_desc(self, P, q, rowptr, colind, val, l, u, colmajor=False):
        """P [B,n,n] or [n,n] (row- or column-major is irrelevant only for symmetric P: pass logical P), q [B,n],
        rowptr int32 [B,m+1] or [m+1], colind int32 [B,nnz_max] or [nnz], val [B,nnz_max] or [nnz], l/u [B,m].
        P may also be a tuple (colptr int32 [B,n+1] or [n+1], rowind int32 [B,pnnz] or [pnnz], val [B,pnnz] or [pnnz]): the full
        symmetric matrix in compressed-column form (sqph_csc_P, the sqph_*_csr_sp entry points)."""
        n, m = self.n, self.m
        sparse_P = P if isinstance(P, tuple) else None
        items = {}
        dev = None
        batch = None

        def prep_idx(a, inner):
            if _is_torch(a):
                import torch

                if a.dtype != torch.int32:
                    raise TypeError("CSR index tensors must be int32")
                if a.is_cuda:
                    a = a.contiguous()
                    shared = a.dim() == 1
                    return a, a.data_ptr(), 0 if shared else a.shape[-1], True, None if shared else a.shape[0]
                a = a.numpy()
            a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
            shared = a.ndim == 1
            return a, a.ctypes.data, 0 if shared else a.shape[-1], False, None if shared else a.shape[0]

        def prep_val(a):
            if _is_torch(a) and a.is_cuda:
                import torch

                tdt = torch.float32 if self.dtype == np.float32 else torch.float64
                if a.dtype != tdt:
                    raise TypeError("tensor dtype %s does not match solver dtype %s" % (a.dtype, self.dtype))
                a = a.contiguous()
                shared = a.dim() == 1
                return a, a.data_ptr(), 0 if shared else a.shape[-1], True, None if shared else a.shape[0]
            if _is_torch(a):
                a = a.numpy()
            a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype))
            shared = a.ndim == 1
            return a, a.ctypes.data, 0 if shared else a.shape[-1], False, None if shared else a.shape[0]

        dense = (("q", q, (n,)), ("l", l, (m,)), ("u", u, (m,)))
        if sparse_P is None:
            dense = (("P", P, (n, n)),) + dense
        for name, arr, shp in dense:
            if tuple(arr.shape[-len(shp):]) != tuple(shp):
                raise ValueError("%s has shape %s, expected [...,%s]" % (name, tuple(arr.shape), shp))
            items[name] = self._prep_one(arr, shp, colmajor)  # colmajor: P is already per-QP column-major (no transposing copy)
        items["rowptr"] = prep_idx(rowptr, m + 1)
        items["colind"] = prep_idx(colind, None)
        items["val"] = prep_val(val)
        if items["rowptr"][0].shape[-1] != m + 1:
            raise ValueError("rowptr must have m+1 entries per QP")
        if sparse_P is not None:
            if len(sparse_P) != 3:
                raise ValueError("sparse P is (colptr, rowind, val)")
            items["P_colptr"] = prep_idx(sparse_P[0], n + 1)
            items["P_rowind"] = prep_idx(sparse_P[1], None)
            items["P_val"] = prep_val(sparse_P[2])
            if items["P_colptr"][0].shape[-1] != n + 1:
                raise ValueError("P colptr must have n+1 entries per QP")
            if int(items["P_rowind"][0].shape[-1]) != int(items["P_val"][0].shape[-1]):
                raise ValueError("P rowind and val must have the same per-QP length")
        for name, it in items.items():
            if dev is None:
                dev = it[3]
            elif dev != it[3]:
                raise ValueError("mixing host and device problem arrays is not supported")
            if it[4] is not None:
                if batch is None:
                    batch = it[4]
                elif batch != it[4]:
                    raise ValueError("inconsistent batch sizes")
        if batch is None:
            batch = self.batch
        if batch > self.batch:
            raise ValueError("batch %d exceeds capacity %d" % (batch, self.batch))
        d = _capi.CsrBatch()
        d.batch = batch
        d.memspace = _capi.DEVICE if dev else _capi.HOST
        for name in ("q", "l", "u") if sparse_P is not None else ("P", "q", "l", "u"):
            setattr(d, name, items[name][1])
            setattr(d, "stride_" + name, items[name][2])
        sp = None
        if sparse_P is not None:
            sp = _capi.CscP()
            sp.colptr, sp.stride_colptr = items["P_colptr"][1], items["P_colptr"][2]
            sp.rowind, sp.stride_rowind = items["P_rowind"][1], items["P_rowind"][2]
            sp.val, sp.stride_val = items["P_val"][1], items["P_val"][2]
            sp.nnz_max = int(items["P_rowind"][0].shape[-1])
        self._sparse_P = sp
        d.A_rowptr, d.stride_rowptr = items["rowptr"][1], items["rowptr"][2]
        d.A_colind, d.stride_colind = items["colind"][1], items["colind"][2]
        d.A_val, d.stride_val = items["val"][1], items["val"][2]
        d.nnz_max = int(items["colind"][0].shape[-1])
        if int(items["val"][0].shape[-1]) != d.nnz_max:
            raise ValueError("colind and val must have the same per-QP length")
        self._keep = [items[k][0] for k in items]
        if dev:
            import torch

            self.set_stream(torch.cuda.current_stream().cuda_stream)
        self._last_batch = batch
        return d

    def _call_csr(self, fn, what, P, q, rowptr, colind, val, l, u, colmajor=False):
        self._push_settings()
        d = self._csr_desc(P, q, rowptr, colind, val, l, u, colmajor)
        if self._sparse_P is not None:  # P given as (colptr, rowind, val): the _sp twin of the entry point
            what += "_sp"
            self._check(getattr(self._L, what)(self._h, ctypes.byref(d), ctypes.byref(self._sparse_P)), what)
            return
        self._check(fn(self._h, ctypes.byref(d)), what)

    def setup_csr(self, P, q, rowptr, colind, val, l, u, colmajor=False):
        self._call_csr(self._L.sqph_setup_csr, "sqph_setup_csr", P, q, rowptr, colind, val, l, u, colmajor)

    def update_qp_csr(self, P, q, rowptr, colind, val, l, u, colmajor=False):
        self._call_csr(self._L.sqph_update_qp_csr, "sqph_update_qp_csr", P, q, rowptr, colind, val, l, u, colmajor)

    def solve_csr(self, P, q, rowptr, colind, val, l, u, colmajor=False):
        self._call_csr(self._L.sqph_solve_csr, "sqph_solve_csr", P, q, rowptr, colind, val, l, u, colmajor)

    def setup_solve_csr(self, P, q, rowptr, colind, val, l, u, colmajor=False):
        self._call_csr(self._L.sqph_setup_solve_csr, "sqph_setup_solve_csr", P, q, rowptr, colind, val, l, u, colmajor)

    def setup_solve_reuse_csr(self, P, q, rowptr, colind, val, l, u, colmajor=False):
        """setup()+solve() with the P, A of the previous set-up (sqph_setup_solve_reuse_csr: the SOC re-solve of a sparse subproblem)."""
        self._call_csr(self._L.sqph_setup_solve_reuse_csr, "sqph_setup_solve_reuse_csr", P, q, rowptr, colind, val, l, u, colmajor)

    def update_solve_csr(self, P, q, rowptr, colind, val, l, u, colmajor=False):
        """update_qp(); solve() in one launch with the iterates kept (sqph_update_solve_csr): successive sparse QPs, warm-started."""
        self._call_csr(self._L.sqph_update_solve_csr, "sqph_update_solve_csr", P, q, rowptr, colind, val, l, u, colmajor)

    def _fetch(self, want):
        B = self._last_batch or self.batch
        out = {}
        if "x" in want:
            out["x"] = np.empty((B, self.n), dtype=self.dtype)
        if "y" in want:
            out["y"] = np.empty((B, self.m), dtype=self.dtype)
        if "z" in want:
            out["z"] = np.empty((B, self.m), dtype=self.dtype)
        if "info" in want:
            out["info"] = np.empty(B, dtype=_capi.INFO_DTYPE)
        p = lambda k: out[k].ctypes.data if k in out else None  # noqa: E731
        self._check(self._L.sqph_get_solution(self._h, B, _capi.HOST, p("x"), p("y"), p("z"), p("info")), "sqph_get_solution")
        return out

    def primal_solution(self):
        return self._fetch("x")["x"]

    def dual_solution(self):
        return self._fetch("y")["y"]

    def z(self):
        return self._fetch("z")["z"]

    def info(self):
        return self._fetch(("info",))["info"].view(np.recarray)

    def solution(self):
        """(x, y, z, info) in one device->host round trip."""
        o = self._fetch(("x", "y", "z", "info"))
        return o["x"], o["y"], o["z"], o["info"].view(np.recarray)

    def set_state(self, x=None, z=None, y=None):
        """Warm-start injection (the reference exposes x, y through non-const accessors)."""
        B = self._last_batch or self.batch

        def cv(a, k):
            if a is None:
                return None, None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=self.dtype), (B, k)))
            return a, a.ctypes.data

        kx, px = cv(x, self.n)
        kz, pz = cv(z, self.m)
        ky, py = cv(y, self.m)
        self._check(self._L.sqph_set_state(self._h, B, _capi.HOST, px, pz, py), "sqph_set_state")

    # ------------------------------------------------------------------ device-side access / timing
    def device_state_ptrs(self):
        x, y, z, info = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._L.sqph_device_state(self._h, ctypes.byref(x), ctypes.byref(y), ctypes.byref(z), ctypes.byref(info)), "sqph_device_state")
        return x.value, y.value, z.value, info.value

    def synchronize(self):
        self._check(self._L.sqph_synchronize(self._h), "sqph_synchronize")

    def enable_timing(self, on=True):
        self._check(self._L.sqph_enable_timing(self._h, 1 if on else 0), "sqph_enable_timing")

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self._check(self._L.sqph_last_kernel_ms(self._h, ctypes.byref(ms)), "sqph_last_kernel_ms")
        return float(ms.value)

    def collect_kernel_ms(self, cap=65536):
        buf = (ctypes.c_float * cap)()
        cnt = ctypes.c_int()
        self._check(self._L.sqph_collect_kernel_ms(self._h, buf, cap, ctypes.byref(cnt)), "sqph_collect_kernel_ms")
        return [float(buf[i]) for i in range(min(cnt.value, cap))]

    def kernel_name(self):
        return self._L.sqph_kernel_name(self._h).decode()

    def algorithmic_bytes_per_qp(self):
        return int(self._L.sqph_algorithmic_bytes(self.n, self.m, self._dt))
