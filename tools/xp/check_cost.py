"""What a termination check costs on the C3 shard (8,192 x (50,100)): 100 iterations with (a) no checks, (b) a check every 25 / 10
iterations that can never pass (eps = 0: the primal-first path), (c) the same with the dual half forced (adaptive rho evaluated at
every check, tolerance so large that rho never changes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch_torch
dev = torch.device("cuda", 0)
B, n, m = 8192, 50, 100
P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=20250231, dtype=torch.float64, device=dev)
def run(name, **st):
    s = QPSolverBatch(n, m, B, device=0)
    for k, v in st.items(): setattr(s.settings, k, v)
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): s.setup_solve(P, q, A, l, u, colmajor=True)
    torch.cuda.synchronize(); s.enable_timing(True)
    for _ in range(20): s.setup_solve(P, q, A, l, u, colmajor=True)
    torch.cuda.synchronize()
    ms = float(np.mean(s.collect_kernel_ms()[-20:]))
    info = s.info()
    print("%-46s %.3f ms   iters/QP %.1f  kernel %s" % (name, ms, np.minimum(info.iter, s.settings.max_iter).mean(), s.kernel_name()))
    s.close(); return ms
base = run("100 iterations, no checks (no-check kernel)", max_iter=100, check_termination=0)
c25 = run("check every 25, eps = 0 (primal-first)", max_iter=100, check_termination=25, eps_abs=0.0, eps_rel=0.0)
c10 = run("check every 10, eps = 0 (primal-first)", max_iter=100, check_termination=10, eps_abs=0.0, eps_rel=0.0)
f25 = run("check every 25 + dual half forced (adaptive)", max_iter=100, check_termination=25, eps_abs=0.0, eps_rel=0.0, adaptive_rho=1, adaptive_rho_interval=25, adaptive_rho_tolerance=1e30)
f10 = run("check every 10 + dual half forced (adaptive)", max_iter=100, check_termination=10, eps_abs=0.0, eps_rel=0.0, adaptive_rho=1, adaptive_rho_interval=10, adaptive_rho_tolerance=1e30)
print("checking kernel without checks firing costs %.3f ms over the no-check kernel (c25 - 4 cheap checks)" % (c25 - base))
print("cheap check  ~ %.1f us per 8,192-QP batch" % ((c10 - c25) / 6 * 1e3))
print("full check   ~ %.1f us per 8,192-QP batch" % ((f10 - f25) / 6 * 1e3))
