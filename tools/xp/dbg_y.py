"""largest y differences against the oracle at one shape — tools/xp/dbg_y.py n m seed"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
n, m, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P, q, A, l, u = random_qp_batch(3, n, m, seed=seed)
s = QPSolverBatch(n, m, 3)
s.settings.adaptive_rho = 1
s.setup_solve(P, q, A, l, u)
x, y, z, info = s.solution()
xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), nthreads=0)
print(s.kernel_name(), "iters", info.iter, io["iter"], "rho updates", info.rho_updates, io["rho_updates"])
print("relerr x %.2e y(relerr1) %.2e z %.2e" % (cases.relerr(x, xo), cases.relerr1(y, yo), cases.relerr(z, zo)))
for b in range(3):
    d = np.abs(y[b] - yo[b]); k = np.argsort(-d)[:4]
    print(" qp", b, "max|y|", np.abs(yo[b]).max(), "worst rows", [(int(i), float(y[b][i]), float(yo[b][i])) for i in k])
    print("   res_prim %.3e / %.3e  res_dual %.3e / %.3e" % (info.res_prim[b], io["res_prim"][b], info.res_dual[b], io["res_dual"][b]))
