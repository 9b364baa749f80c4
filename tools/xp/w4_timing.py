"""four-wave 16 x 16 grid (56 < n <= 64, m <= 128): parity on a few QPs, then kernel time of 4,096 QPs in the three modes — tools/xp/w4_timing.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch, random_qp_batch_torch
for (n, m, b) in ((60, 120, 6), (64, 128, 6), (57, 3, 4)):
    P, q, A, l, u = random_qp_batch(b, n, m, seed=9)
    for kw in (dict(max_iter=80, check_termination=0), dict(adaptive_rho=1)):
        s = QPSolverBatch(n, m, b)
        for k, v in kw.items(): setattr(s.settings, k, v)
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), nthreads=0)
        print(n, m, kw, s.kernel_name(), "x %.2e y %.2e" % (cases.relerr(x, xo), cases.relerr1(y, yo)), "status/iter equal", bool((info.status == io["status"]).all() and (info.iter == io["iter"]).all()))
for (n, m) in ((60, 120), (64, 128)):
    B = 4096
    P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=5, dtype=torch.float64, device=torch.device("cuda:0"))
    for name, kw in (("fixed200", dict(max_iter=200, check_termination=0)), ("default", dict()), ("fixed10", dict(max_iter=10, check_termination=0))):
        s = QPSolverBatch(n, m, B)
        for k, v in kw.items(): setattr(s.settings, k, v)
        s.setup_solve(P, q, A, l, u, colmajor=True)
        torch.cuda.synchronize()
        s.enable_timing(True)
        for _ in range(5): s.setup_solve(P, q, A, l, u, colmajor=True)
        torch.cuda.synchronize()
        print("4096 x (%d,%d) %s:" % (n, m, name), s.kernel_name(), "kernel ms %.4f" % np.mean(s.collect_kernel_ms()))
