"""Debug helper (GPU): per-phase s_memtime ticks of the fused call's set-up part (wave 0 of every QP, mean over the batch).
Needs a library built with -DSQPH_SETUP_TIMING:  tools/slim_build.sh sqp_solver_amd/lib/libsqp_hip_slimsetup.so -DSQPH_SETUP_TIMING
then  SQPH_LIB=$PWD/sqp_solver_amd/lib/libsqp_hip_slimsetup.so python tools/setup_timing.py 50 100 8192"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
n, m, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P, q, A, l, u = random_qp_batch(64, n, m, seed=1)
rep = (B + 63) // 64
tile = lambda a: np.concatenate([a] * rep)[:B]
s = QPSolverBatch(n, m, B)
s.settings.max_iter = 1
s.settings.check_termination = 0
args = [tile(a) for a in (P, q, A, l, u)]
s.setup_solve(*args)
s.setup_solve(*args)
x = s.solution()[0]
names = ["load_A", "S=A'RA", "P+scale", "pivots", "W_scale(+store)", "build_B", "W'_tile", "other"]
t = x[:, :8].mean(axis=0)
print(s.kernel_name(), "set-up ticks (wave 0):", " ".join("%s=%.0f" % (a, v) for a, v in zip(names, t)), "total=%.0f" % t.sum())
