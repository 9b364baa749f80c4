"""CPU tests of the oracle itself: it must reproduce every known answer the reference's own tests
hold for the QP path and agree with independent numpy/scipy restatements."""
import numpy as np
import pytest

import cases
import golden_io
import oracle
from sqp_solver_amd.problems import SIMPLE_QP as S, random_qp_batch


def make_oracle(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    """Adapter: the oracle's single-QP class behind the batched test surface."""

    class _B:
        def __init__(self):
            self.s = [oracle.QPSolver(dtype, legacy=legacy_cold_start) for _ in range(batch)]
            self.settings = self.s[0].settings
            self.n, self.m = n, m

        def _each(self, fn, P, q, A, l, u):
            pick = lambda a, b, nd: None if a is None else (a if np.asarray(a).ndim == nd else a[b])  # noqa: E731
            self._B = np.asarray(q).shape[0] if np.asarray(q).ndim == 2 else batch
            for b in range(self._B):
                st = self.s[b].settings
                for k, _ in self.settings._fields_:
                    setattr(st, k, getattr(self.settings, k))
                Ab = pick(A, b, 2) if m else np.zeros((0, n))
                lb = pick(l, b, 1) if m else np.zeros(0)
                ub = pick(u, b, 1) if m else np.zeros(0)
                getattr(self.s[b], fn)(pick(P, b, 2), pick(q, b, 1), Ab, lb, ub)

        def setup(self, *qp):
            self._each("setup", *qp)

        def update_qp(self, *qp):
            self._each("update_qp", *qp)

        def solve(self, *qp):
            self._each("solve", *qp)

        def setup_solve(self, *qp):
            self._each("setup", *qp)
            self._each("solve", *qp)

        def set_state(self, x=None, z=None, y=None):
            for b, s in enumerate(self.s):
                s.set_state(None if x is None else x[b], None if z is None else z[b], None if y is None else y[b])

        def solution(self):
            B = self._B
            x = np.stack([self.s[b].primal_solution() if self.s[b].n else np.zeros(n, dtype) for b in range(B)])
            y = np.stack([self.s[b].dual_solution() if self.s[b].n else np.zeros(m, dtype) for b in range(B)])
            z = np.stack([self.s[b].z() if self.s[b].n else np.zeros(m, dtype) for b in range(B)])
            info = np.zeros(B, oracle.INFO_DTYPE)
            for b in range(B):
                i = self.s[b].info
                info[b] = (i.status, i.iter, i.rho_updates, 0, i.rho_estimate, i.res_prim, i.res_dual)
            return x, y, z, info.view(np.recarray)

    return _B()


@pytest.mark.parametrize("case", cases.REFERENCE_CASES, ids=lambda f: f.__name__)
def test_reference_known_answers(case):
    """tests/qp_solver_test.cpp, tests/unsupported/qp_solver_test.cpp, tests/qp_solver_sparse_test.cpp"""
    case(make_oracle)


def test_TestConstraint_table():
    """tests/qp_solver_test.cpp:127-156 — static constr_type_init on the 5 (l,u) pairs incl. +-10*1e16"""
    T = 1e16
    l = [-10 * T, -1, -10 * T, -3, 42]
    u = [10 * T, 10 * T, 2, 4, 42]
    expect = [oracle.LOOSE_BOUNDS, oracle.INEQUALITY_CONSTRAINT, oracle.INEQUALITY_CONSTRAINT,
              oracle.INEQUALITY_CONSTRAINT, oracle.EQUALITY_CONSTRAINT]
    assert list(oracle.constr_type_init(l, u)) == expect
    assert list(oracle.constr_type_init(l, u, np.float32)) == expect
    # the product's host utility (C-ABI, no device needed) implements the same table
    from sqp_solver_amd import constr_type_init

    assert list(constr_type_init(l, u)) == expect
    assert list(constr_type_init(l, u, np.float32)) == expect
    assert list(constr_type_init([-np.inf, 0.0, 0.0], [np.inf, np.inf, 5e-5])) == [2, 0, 1]


def test_analytic_solution_of_fixture():
    """x*=[0.3,0.7], y*=[-2.9,0,0.2] from P x + q + A'y = 0 with row 0 an equality, row 2 active-upper."""
    s = oracle.QPSolver()
    s.settings.eps_abs = s.settings.eps_rel = 1e-9
    s.settings.max_iter = 20000
    s.setup(S["P"], S["q"], S["A"], S["l"], S["u"])
    s.solve(S["P"], S["q"], S["A"], S["l"], S["u"])
    assert s.info.status == oracle.SOLVED
    assert np.allclose(s.primal_solution(), S["solution"], atol=1e-7)
    assert np.allclose(s.dual_solution(), S["dual"], atol=1e-6)


def test_status_bookkeeping():
    """iter == max_iter+1 on exhaustion; rho_updates accumulates across setup()."""
    s = oracle.QPSolver()
    assert s.info.status == oracle.UNINITIALIZED
    s.solve(S["P"], S["q"], S["A"], S["l"], S["u"]) if False else None
    s.settings.max_iter = 10
    s.setup(S["P"], S["q"], S["A"], S["l"], S["u"])
    assert s.info.status == oracle.UNSOLVED and s.info.rho_updates == 1
    s.solve(S["P"], S["q"], S["A"], S["l"], S["u"])
    assert s.info.status == oracle.MAX_ITER_EXCEEDED and s.info.iter == 11
    s.setup(S["P"], S["q"], S["A"], S["l"], S["u"])
    assert s.info.rho_updates == 2


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 2e-3)])
def test_ldlt_against_numpy(dtype, tol):
    """the pivoted LDL^T restatement: P K P' = L D L' and K^-1 b, on quasi-definite KKT matrices"""
    rng = np.random.default_rng(0)
    for (n, m) in ((2, 3), (7, 11), (20, 40)):
        P, q, A, l, u = random_qp_batch(1, n, m, seed=n)
        rho = np.where(rng.uniform(size=m) < 0.2, 100.0, 0.1)
        K = np.block([[P[0] + 1e-6 * np.eye(n), A[0].T], [A[0], -np.diag(1 / rho)]])
        b = rng.standard_normal(n + m)
        ok, LD, tr, sol = oracle.ldlt_factor_solve(K, b, dtype)
        assert ok
        ref = np.linalg.solve(K, b)
        assert np.max(np.abs(sol - ref)) / np.max(np.abs(ref)) < tol
        N = n + m
        perm = np.arange(N)
        for k in range(N):
            perm[[k, tr[k]]] = perm[[tr[k], k]]
        L = np.tril(LD.astype(np.float64), -1) + np.eye(N)
        D = np.diag(np.diag(LD).astype(np.float64))
        Kp = K[np.ix_(perm, perm)]
        assert np.max(np.abs(L @ D @ L.T - Kp)) / np.max(np.abs(K)) < tol
        # Eigen's pivot rule: the order is by |original diagonal|, largest first (ties: first)
        d0 = np.abs(np.diag(K))[perm]
        assert (np.diff(d0) <= 1e-12 * d0.max()).all()


def numpy_admm(P, q, A, l, u, iters, rho=0.1, sigma=1e-6, alpha=1.0):
    """Independent restatement of one QP's iteration (full KKT, numpy.linalg.solve)."""
    n, m = P.shape[0], A.shape[0]
    ct = np.where((l < -1e16) & (u > 1e16), 2, np.where(u - l < 1e-4, 1, 0))
    rv = np.where(ct == 2, 1e-6, np.where(ct == 1, 1e3 * rho, rho))
    K = np.block([[P + sigma * np.eye(n), A.T], [A, -np.diag(1 / rv)]])
    x, z, y = np.zeros(n), np.zeros(m), np.zeros(m)
    for _ in range(iters):
        sol = np.linalg.solve(K, np.concatenate([sigma * x - q, z - y / rv]))
        xt, nu = sol[:n], sol[n:]
        zt = z + (nu - y) / rv
        x = alpha * xt + (1 - alpha) * x
        zr = alpha * zt + (1 - alpha) * z
        zn = np.minimum(np.maximum(zr + y / rv, l), u)
        y = y + rv * (zr - zn)
        z = zn
    return x, y, z


@pytest.mark.parametrize("n,m,alpha", [(5, 8, 1.0), (20, 40, 1.0), (20, 40, 1.6), (50, 100, 1.0)])
def test_oracle_vs_independent_numpy(n, m, alpha):
    P, q, A, l, u = random_qp_batch(2, n, m, seed=42)
    st = oracle.default_settings(max_iter=80, check_termination=0, alpha=alpha)
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, st)
    for b in range(2):
        x, y, z = numpy_admm(P[b], q[b], A[b], l[b], u[b], 80, alpha=alpha)
        assert np.max(np.abs(x - xo[b])) / np.max(np.abs(x)) < 1e-9
        assert np.max(np.abs(y - yo[b])) / np.max(np.abs(y)) < 1e-8


def test_golden_fixtures_match_oracle():
    """the committed vectors (tests/golden, made by make_golden.py) are what today's oracle build produces"""
    gs = golden_io.load_all()
    assert len(gs) >= 7
    for name, g in gs:
        st = oracle.default_settings()
        golden_io.apply_settings(st, g)
        x, y, z, info = oracle.solve_batch(g["P"], g["q"], g["A"], g["l"], g["u"], st, nthreads=2)
        assert (info["status"] == g["status"]).all() and (info["iter"] == g["iter"]).all(), name
        assert np.allclose(x, g["x"], rtol=1e-9, atol=1e-12) and np.allclose(y, g["y"], rtol=1e-8, atol=1e-10), name


def test_batch_driver_threads_agree():
    P, q, A, l, u = random_qp_batch(24, 10, 17, seed=1)
    a = oracle.solve_batch(P, q, A, l, u, nthreads=1)
    b = oracle.solve_batch(P, q, A, l, u, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
