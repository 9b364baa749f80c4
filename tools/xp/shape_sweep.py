"""fixed-200 kernel time over a grid of dense shapes, with the fraction of the fp64 peak the useful multiply-adds reach — tools/xp/shape_sweep.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch_torch
B = 2048
print("   n    m  kernel                      ms/2048 QPs   us/QP   fp64 frac (2 (2 m n + n^2) flops per iteration)")
for n in (8, 12, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112):
    for m in sorted(set((max(1, n // 2), n, 2 * n, 4 * n))):
        if m > 448 or (n > 56 and m > 224): continue
        P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=5, dtype=torch.float64, device=torch.device("cuda:0"))
        s = QPSolverBatch(n, m, B)
        s.settings.max_iter, s.settings.check_termination = 200, 0
        s.setup_solve(P, q, A, l, u, colmajor=True)
        torch.cuda.synchronize()
        s.enable_timing(True)
        for _ in range(4): s.setup_solve(P, q, A, l, u, colmajor=True)
        torch.cuda.synchronize()
        ms = float(np.mean(s.collect_kernel_ms()))
        fl = 2.0 * (2 * m * n + n * n) * 200 * B
        print("%4d %4d  %-26s %9.4f  %7.3f   %5.1f %%" % (n, m, s.kernel_name(), ms, ms * 1e3 / B, 100 * fl / (ms * 1e-3) / 78.6e12))
