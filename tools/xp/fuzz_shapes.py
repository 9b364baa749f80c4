"""one-off parity fuzz over random dense shapes (every register-tiled grid, the CU-wide dense kernel, the lane kernels): fixed iterations and
default termination against the oracle on batches of 3 — tools/xp/fuzz_shapes.py [count] [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
count = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = {}
bad = 0
for it in range(count):
    n = int(rng.integers(1, 113)); m = int(rng.integers(0, 449)) if rng.random() < 0.7 else int(rng.integers(0, 2 * n + 1))
    if n > 56 and m > 208: m = int(rng.integers(0, 209))
    b = 3
    P, q, A, l, u = random_qp_batch(b, n, max(m, 1), seed=1000 + it)
    if m == 0: continue
    for kw in (dict(max_iter=60, check_termination=0), dict(adaptive_rho=1)):
        s = QPSolverBatch(n, m, b)
        for k, v in kw.items(): setattr(s.settings, k, v)
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), nthreads=0)
        ex, ey = cases.relerr(x, xo), cases.relerr1(y, yo)
        same = bool((info.status == io["status"]).all() and (info.iter == io["iter"]).all())
        k = s.kernel_name()
        w = worst.setdefault(k, [0.0, 0.0, 0, 0])
        w[0] = max(w[0], ex); w[1] = max(w[1], ey); w[2] += 1; w[3] += 0 if same else 1
        if not (ex < 1e-8 and ey < 1e-7) or (not same and "max_iter" in kw):
            bad += 1
            print("BAD", n, m, kw, k, "x %.2e y %.2e" % (ex, ey), "status/iter equal", same)
for k, w in sorted(worst.items()): print("%-28s runs %3d  max x %.2e  max y %.2e  status/iter differences %d" % (k, w[2], w[0], w[1], w[3]))
print("bad", bad)
