"""dense problems beyond the register-tiled shapes (the CU-wide kernel's dense-A mode, csr_dense.hip): parity against the oracle on a
few QPs, then the time of 512 x (200,400) — tools/xp/dense_large.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch, random_qp_batch_torch
for (n, m, b) in ((200, 400, 4), (120, 60, 4), (50, 500, 4), (224, 512, 2), (113, 1, 3)):
    P, q, A, l, u = random_qp_batch(b, n, m, seed=9)
    for kw in (dict(max_iter=80, check_termination=0), dict(adaptive_rho=1)):
        s = QPSolverBatch(n, m, b)
        for k, v in kw.items(): setattr(s.settings, k, v)
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), nthreads=0)
        print(n, m, kw, s.kernel_name(), "x %.2e y %.2e" % (cases.relerr(x, xo), cases.relerr1(y, yo)), "status/iter equal", bool((info.status == io["status"]).all() and (info.iter == io["iter"]).all()))
for force in (False, True):
    n, m, B = 200, 400, 512
    P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=5, dtype=torch.float64, device=torch.device("cuda:0"))
    s = QPSolverBatch(n, m, B, force_generic=force)
    s.settings.max_iter, s.settings.check_termination = 200, 0
    s.setup_solve(P, q, A, l, u, colmajor=True)
    torch.cuda.synchronize()
    s.enable_timing(True)
    for _ in range(3): s.setup_solve(P, q, A, l, u, colmajor=True)
    torch.cuda.synchronize()
    print("512 x (200,400) dense, 200 iterations:", s.kernel_name(), "kernel ms", np.mean(s.collect_kernel_ms()))
