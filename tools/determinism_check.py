import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch_torch
def run(n, m, B, iters, reps, mode):
    P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=5, device="cuda:0")
    s = QPSolverBatch(n, m, B, device=0)
    if mode == "fixed":
        s.settings.max_iter = iters; s.settings.check_termination = 0
    elif mode == "adaptive":
        s.settings.adaptive_rho = 1; s.settings.adaptive_rho_interval = 25
    ref = None; bad = 0
    for r in range(reps):
        s.setup_solve(P, q, A, l, u, colmajor=True)
        x, y, z, info = s.solution()
        cur = (x.copy(), y.copy(), z.copy(), info.iter.copy(), info.status.copy())
        if ref is None: ref = cur
        else:
            same = all(np.array_equal(a, b) for a, b in zip(ref, cur))
            bad += (not same)
    print(n, m, B, mode, s.kernel_name(), "reps", reps, "runs differing from the first:", bad)
run(50, 100, 8192, 200, 40, "fixed")
run(50, 100, 8192, 0, 40, "default")
run(50, 100, 8192, 0, 30, "adaptive")
run(20, 40, 16384, 200, 40, "fixed")
run(20, 40, 16384, 0, 40, "default")
run(60, 120, 2048, 100, 30, "fixed")
run(100, 200, 1024, 100, 20, "fixed")
run(30, 60, 8192, 100, 30, "default")
run(8, 12, 65536, 100, 30, "default")
run(2, 3, 65536, 100, 30, "adaptive")
run(250, 300, 512, 0, 12, "adaptive")   # the CU-wide kernel's dense mode at tile edge 8 (round 6), full occupancy
run(240, 500, 512, 100, 12, "fixed")
run(200, 400, 512, 0, 12, "default")
