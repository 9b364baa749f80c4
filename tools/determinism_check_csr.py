"""Run-to-run determinism of the block-row sparse kernel (config 5 shape, full occupancy): repeated fused solves must be bit-identical
(the S phase adds into LDS with ds_add_f64: the additions to one address come from one wavefront in program order)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import bench_csr
from sqp_solver_amd import QPSolverBatch
def run(B, n, m, mode, reps):
    P, q, rp, ci, v, l, u, A, nnz = bench_csr.make(B, n, m, 0.05, 11, torch.device("cuda:0"))
    s = QPSolverBatch(n, m, B, device=0)
    if mode == "fixed":
        s.settings.max_iter = 100; s.settings.check_termination = 0
    else:
        s.settings.adaptive_rho = 1; s.settings.adaptive_rho_interval = 25
    ref = None; bad = 0
    for r in range(reps):
        s.setup_solve_csr(P, q, rp, ci, v, l, u, colmajor=True)
        x, y, z, info = s.solution()
        cur = (x.copy(), y.copy(), z.copy(), info.iter.copy(), info.status.copy())
        if ref is None: ref = cur
        else: bad += not all(np.array_equal(a, b) for a, b in zip(ref, cur))
    print(n, m, B, mode, s.kernel_name(), "reps", reps, "runs differing from the first:", bad)
run(2048, 200, 400, "fixed", 12)
run(1024, 200, 400, "adaptive", 8)
run(2048, 120, 300, "fixed", 8)
