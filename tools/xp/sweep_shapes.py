"""one termination check after 25 iterations on many dense shapes: reported residuals and iterates against the oracle
(tools/xp/sweep_shapes.py lo_n hi_n step_n  m1,m2,...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import lds_poison
print("LDS poison before every call:", lds_poison.install())
lo, hi, st = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ms = [int(v) for v in sys.argv[4].split(",")]
bad = 0
for n in range(lo, hi + 1, st):
    for m in ms:
        P, q, A, l, u = random_qp_batch(2, n, m, seed=5)
        s = QPSolverBatch(n, m, 2)
        s.settings.max_iter, s.settings.check_termination = 25, 25
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, oracle.default_settings(max_iter=25, check_termination=25))
        ex = np.max(np.abs(x - xo)) / np.max(np.abs(xo))
        er = max(np.max(np.abs(info.res_prim - io["res_prim"]) / io["res_prim"]), np.max(np.abs(info.res_dual - io["res_dual"]) / io["res_dual"]))
        ok = ex < 1e-6 and er < 1e-6
        bad += not ok
        print(n, m, s.kernel_name(), "ok" if ok else "FAIL", "%.1e %.1e" % (ex, er), flush=True)
        s.close()
print("failures:", bad)
