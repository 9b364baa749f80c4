"""mean iterations and rho updates per QP of the C3 shard under the default and the SQP driver's settings — tools/xp/rho_updates.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch_torch
n, m, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 100, 8192)
P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=0, dtype=torch.float64, device=torch.device("cuda:0"))
for mode in ("default", "sqp"):
    s = QPSolverBatch(n, m, B)
    bench.apply_mode(s.settings, mode, 200)
    s.setup_solve(P, q, A, l, u, colmajor=True)
    x, y, z, info = s.solution()
    it = np.asarray(info.iter); ru = np.asarray(info.rho_updates)
    print(mode, "iters mean %.1f" % it.mean(), "hist", np.bincount(it // 5)[:45].tolist() if mode == "sqp" else np.bincount(it // 25).tolist(), "rho updates mean %.2f" % ru.mean(), "hist", np.bincount(ru).tolist())
