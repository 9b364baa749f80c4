import sys, time, os
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/sqp_solver_amd') else '.')
import numpy as np
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
for (n, m, B) in ((2, 3, 1024), (2, 3, 64), (3, 3, 256)):
    P, q, A, l, u = random_qp_batch(B, n, m, seed=5)
    s = QPSolverBatch(n, m, B)
    st = s.settings
    st.warm_start, st.check_termination, st.eps_abs, st.eps_rel = 1, 10, 1e-4, 1e-4
    st.max_iter, st.adaptive_rho, st.adaptive_rho_interval, st.alpha = 100, 1, 50, 1.6
    Pc = np.ascontiguousarray(np.swapaxes(P, 1, 2)); Ac = np.ascontiguousarray(np.swapaxes(A, 1, 2))
    for _ in range(5):
        s.setup_solve(Pc, q, Ac, l, u, colmajor=True); s.solution()
    s.enable_timing(True)
    t0 = time.perf_counter()
    R = 50
    for _ in range(R):
        s.setup_solve(Pc, q, Ac, l, u, colmajor=True)
        x, y, z, info = s.solution()
    dt = (time.perf_counter() - t0) / R
    ms = s.collect_kernel_ms()
    print("n=%d m=%d batch=%d kernel %s: wall per call+fetch %.1f us, kernel %.1f us, mean iters %.1f" % (n, m, B, s.kernel_name(), dt * 1e6, np.mean(ms) * 1e3, info.iter.mean()))
