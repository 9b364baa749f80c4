"""Per-iteration cost of the checking instantiations against the no-check ones (GPU): 100 iterations with four checks that never pass
against 100 unchecked iterations, for one shape of every kernel family."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch_torch
for (n, m, B) in ((50, 100, 8192), (20, 40, 4096), (8, 12, 16384), (2, 3, 65536), (100, 100, 2048), (30, 200, 4096), (50, 400, 2048), (100, 200, 2048), (200, 400, 512), (250, 300, 64)):
    P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=3, dtype=torch.float64, device=torch.device("cuda:0"))
    ms = {}
    for name, ct in (("fixed", 0), ("checked", 25)):
        s = QPSolverBatch(n, m, B, device=0)
        s.settings.max_iter = 100; s.settings.check_termination = ct; s.settings.eps_abs = s.settings.eps_rel = 1e-300
        s.setup_solve(P, q, A, l, u, colmajor=True)
        s.enable_timing(True)
        for _ in range(3): s.setup_solve(P, q, A, l, u, colmajor=True)
        ms[name] = float(np.median(s.collect_kernel_ms()[-3:])); k = s.kernel_name(); s.close()
    print("%4d x %4d batch %6d %-22s fixed %8.3f ms  checked %8.3f ms  ratio %.2f" % (n, m, B, k, ms["fixed"], ms["checked"], ms["checked"] / ms["fixed"]))
