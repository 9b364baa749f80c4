import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
B, n, m = 8192, 50, 100
P, q, A, l, u = random_qp_batch(B, n, m, seed=1)
# the C-ABI layout (per-QP column-major) prepared once, outside the timed region
Pc = np.ascontiguousarray(P.transpose(0, 2, 1)); Ac = np.ascontiguousarray(A.transpose(0, 2, 1))
s = QPSolverBatch(n, m, B)
s.settings.max_iter = 200; s.settings.check_termination = 0
for _ in range(2):
    s.setup_solve(Pc, q, Ac, l, u, colmajor=True); s.solution()
t0 = time.perf_counter(); K = 5
for _ in range(K):
    s.setup_solve(Pc, q, Ac, l, u, colmajor=True)
    x, y, z, info = s.solution()
dt = (time.perf_counter() - t0) / K
print("host-memspace round trip: %.2f ms per %d QPs = %.3g QP/s (H2D 508 MB pageable + kernel + D2H)" % (dt * 1e3, B, B / dt))
