"""Timing + parity of dense shapes beyond the register-tiled kernels: tools/xp/big_dense.py n m batch iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
n, m, B, iters = (int(a) for a in sys.argv[1:5])
P, q, A, l, u = random_qp_batch(min(B, 8), n, m, seed=3)
rep = (B + 7) // 8
tile = lambda a: np.concatenate([a] * rep)[:B]
args = [tile(a) for a in (P, q, A, l, u)]
s = QPSolverBatch(n, m, B)
s.settings.max_iter, s.settings.check_termination = iters, 0
s.setup_solve(*args)
s.enable_timing(True)
for _ in range(3):
    s.setup_solve(*args)
ms = s.collect_kernel_ms()
x, y, z, info = s.solution()
k = min(B, 4)
xo, yo, zo, io = oracle.solve_batch(P[:k], q[:k], A[:k], l[:k], u[:k], oracle.default_settings(max_iter=iters, check_termination=0))
rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
print("%d x (%d, %d) %d iters: kernel %s  %.3f ms (min of 3)  rel err x %.2e y %.2e" % (B, n, m, iters, s.kernel_name(), min(ms[-3:]), rel(x[:k], xo), rel(y[:k], yo)))
