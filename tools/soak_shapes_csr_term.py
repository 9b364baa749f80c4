"""One-off soak (GPU): the block-row sparse kernel under termination checks / adaptive rho / the SQP driver's settings over random
sparse shapes, against the CPU oracle (status, iteration counts, rho updates, x and y)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle, cases
from test_gpu_parity import make_gpu
import lds_poison
print("LDS poison before every call:", lds_poison.install())
from sqp_solver_amd.problems import random_csr_qp_batch
rng = np.random.default_rng(4242 + int(os.environ.get("SQPH_SOAK_SEED", "0")))
tot = bad = 0; kern = {}; worst = 0.0; notes = []
for t in range(36):
    n = int(rng.integers(17, 225)); m = int(rng.integers(129, 513)); dens = float(rng.choice([0.03, 0.06, 0.15]))
    if n * m * dens > 6000: dens = 6000.0 / (n * m)
    B = 4
    P, q, rp, ci, v, l, u, A = random_csr_qp_batch(B, n, m, density=max(dens, 1.5 / n), seed=1000 + t)
    mode = t % 3
    s = make_gpu(n, m, B)
    if mode == 1:
        s.settings.adaptive_rho, s.settings.adaptive_rho_interval = 1, 25
    elif mode == 2:  # src/sqp.cpp:15-23
        s.settings.warm_start, s.settings.check_termination, s.settings.eps_abs, s.settings.eps_rel = 1, 10, 1e-4, 1e-4
        s.settings.max_iter, s.settings.adaptive_rho, s.settings.adaptive_rho_interval, s.settings.alpha = 100, 1, 50, 1.6
    s.setup_solve_csr(P, q, rp, ci, v, l, u)
    x, y, z, info = s.solution()
    kern[s.kernel_name()] = kern.get(s.kernel_name(), 0) + 1
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings))
    for b in range(B):
        tot += 1
        same = info.status[b] == io["status"][b] and info.iter[b] == io["iter"][b] and info.rho_updates[b] == io["rho_updates"][b]
        ex = cases.relerr(x[b:b + 1], xo[b:b + 1]); ey = cases.relerr(y[b:b + 1], yo[b:b + 1])
        # the reported dual residual against the oracle's: 1e-5 relative plus the rounding level of its terms (P x + q + A'y with |q| ~ 1:
        # iterates that agree to 1e-11 leave residuals that agree to ~1e-11 absolute, whatever the residual's own size — a converged
        # iterate's residual of 1e-12 is pure noise)
        scale = max(1.0, float(np.max(np.abs(q[b]))))
        rd = abs(info.res_dual[b] - io["res_dual"][b]) / (abs(io["res_dual"][b]) + 1e-5 * scale)
        if same: worst = max(worst, ex, ey)
        if not same or ex > 1e-6 or ey > 1e-6 or rd > 1e-5:
            bad += 1
            notes.append((n, m, round(dens, 3), mode, b, int(info.status[b]), int(io["status"][b]), int(info.iter[b]), int(io["iter"][b]), int(info.rho_updates[b]), int(io["rho_updates"][b]), "%.1e %.1e rd %.1e (res_dual %.2e, oracle %.2e)" % (ex, ey, rd, info.res_dual[b], io["res_dual"][b])))
    s.close()
print("kernels:", sorted(kern.items()))
print("QPs %d, differing in status / iterations / rho updates or beyond 1e-6 (res_dual: 1e-5 relative + 1e-10 absolute): %d; worst x / y error among the equal ones %.2e" % (tot, bad, worst))
for r in notes[:12]: print(r)
