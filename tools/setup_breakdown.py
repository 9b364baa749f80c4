#!/usr/bin/env python
"""Where the fused call's time goes on the C3 shard: setup() alone (load, S, factor), setup+solve with max_iter = 0 (adds B = A W'
and the W' tile), and with 200 iterations.  Device-resident inputs, HIP-event kernel times (median of 10)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch_torch

n, m, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (50, 100, 8192)))
P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=1, device="cuda:0")
s = QPSolverBatch(n, m, B, device=0)
s.set_stream(torch.cuda.current_stream().cuda_stream)
s.settings.check_termination = 0


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s.enable_timing(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = s.collect_kernel_ms()
    s.enable_timing(False)
    return float(np.median(ms))


t_setup = timed(lambda: s.setup(P, q, A, l, u, colmajor=True))
out = {"kernel": s.kernel_name(), "setup_only_ms": t_setup}
for it in (0, 1, 50, 100, 200):
    s.settings.max_iter = it
    out["fused_%d_ms" % it] = timed(lambda: s.setup_solve(P, q, A, l, u, colmajor=True))
print(out)
print("per-iteration ms: %.5f" % ((out["fused_200_ms"] - out["fused_100_ms"]) / 100))
# cost of one termination check: 100 iterations with 4 checks that never pass (eps = 0) against 100 iterations without
s.settings.max_iter = 100
s.settings.check_termination = 25
s.settings.eps_abs = s.settings.eps_rel = 0.0
t_chk = timed(lambda: s.setup_solve(P, q, A, l, u, colmajor=True))
print("100 iterations with 4 checks: %.4f ms; without: %.4f ms; one check = %.4f ms = %.1f iterations" % (
    t_chk, out["fused_100_ms"], (t_chk - out["fused_100_ms"]) / 4, (t_chk - out["fused_100_ms"]) / 4 / ((out["fused_200_ms"] - out["fused_100_ms"]) / 100)))
