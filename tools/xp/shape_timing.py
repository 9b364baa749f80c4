"""kernel time of dense batches at given shapes, fixed-200 / default / fixed-10, with a parity probe on a few QPs — tools/xp/shape_timing.py n,m,batch ..."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch, random_qp_batch_torch
for spec in sys.argv[1:]:
    n, m, B = [int(x) for x in spec.split(",")]
    P, q, A, l, u = random_qp_batch(4, n, m, seed=9)
    s = QPSolverBatch(n, m, 4)
    s.settings.adaptive_rho = 1
    s.setup_solve(P, q, A, l, u)
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), nthreads=0)
    print(n, m, s.kernel_name(), "x %.2e y %.2e" % (cases.relerr(x, xo), cases.relerr1(y, yo)), "status/iter equal", bool((info.status == io["status"]).all() and (info.iter == io["iter"]).all()))
    P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=5, dtype=torch.float64, device=torch.device("cuda:0"))
    for name, kw in (("fixed200", dict(max_iter=200, check_termination=0)), ("default", dict()), ("fixed10", dict(max_iter=10, check_termination=0))):
        s = QPSolverBatch(n, m, B)
        for k, v in kw.items(): setattr(s.settings, k, v)
        s.setup_solve(P, q, A, l, u, colmajor=True)
        torch.cuda.synchronize()
        s.enable_timing(True)
        for _ in range(5): s.setup_solve(P, q, A, l, u, colmajor=True)
        torch.cuda.synchronize()
        print("  %d x (%d,%d) %s:" % (B, n, m, name), s.kernel_name(), "kernel ms %.4f" % np.mean(s.collect_kernel_ms()))
