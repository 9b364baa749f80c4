import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
for (n, m, fg) in ((124, 20, 0), (124, 20, 1), (124, 40, 0), (124, 63, 0), (124,64,0), (124, 65, 0), (168, 20, 0), (60, 20, 0)):
    P, q, A, l, u = random_qp_batch(2, n, m, seed=5)
    s = QPSolverBatch(n, m, 2, force_generic=bool(fg))
    s.settings.max_iter, s.settings.check_termination = 25, 25
    s.setup_solve(P, q, A, l, u)
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, oracle.default_settings(max_iter=25, check_termination=25))
    print(n, m, s.kernel_name(), "res_prim", info.res_prim, io["res_prim"], "res_dual", info.res_dual, io["res_dual"], "status", info.status, io["status"], info.iter, io["iter"])
