"""fp32-product variant (SQPH_FLAG_F32_ARITH at the BASELINE dense shapes): errors of x, y, z against the fp64 solution of the same
float-valued problem, next to the errors of the reference's float path (the float oracle) — the ratio is what the tests bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
f64 = lambda a: np.asarray(a, dtype=np.float64)
for (n, m, b) in ((20, 40, 512), (50, 100, 256), (30, 60, 64), (56, 112, 32)):
    for seed in (11, 12):
        P, q, A, l, u = random_qp_batch(b, n, m, seed=seed, dtype=np.float32)
        s = QPSolverBatch(n, m, b, dtype=np.float32, device=0, f32_arith=True)
        s.settings.max_iter, s.settings.check_termination = 200, 0
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        st = cases.oracle_settings(s.settings)
        xo, yo, zo, _ = oracle.solve_batch(P, q, A, l, u, st, dtype=np.float32)
        x64, y64, z64, _ = oracle.solve_batch(f64(P), f64(q), f64(A), f64(l), f64(u), st)
        r = cases.relerr
        print(n, m, seed, s.kernel_name(), "x %.2e (float oracle %.2e)  y %.2e (%.2e)  z %.2e (%.2e)" % (r(x, x64), r(xo, x64), r(y, y64), r(yo, y64), r(z, z64), r(zo, z64)))
