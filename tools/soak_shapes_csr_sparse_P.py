"""One-off soak (GPU): the sparse-P entry points (sqph_*_csr_sp) over random sparse shapes and P densities — fixed iterations, default
termination, adaptive rho, the SQP driver's settings; fused and stateful call sequences — against the dense-P calls on the matrix the
compressed columns encode (bit-identical x, y, z, status, iterations, rho updates, reported residuals) and against the CPU oracle."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle, cases
from test_gpu_parity import make_gpu
import lds_poison
print("LDS poison before every call:", lds_poison.install())
from sqp_solver_amd.problems import random_csr_qp_batch
rng = np.random.default_rng(90210 + int(os.environ.get("SQPH_SOAK_SEED", "0")))
tot = differ = orc_bad = 0; kern = {}; notes = []
for t in range(48):
    n = int(rng.integers(5, 225)); m = int(rng.integers(8, 513)); dens = float(rng.choice([0.03, 0.06, 0.15]))
    if n * m * dens > 6000: dens = 6000.0 / (n * m)
    pd = float(rng.choice([0.0, 0.02, 0.1, 0.5]))
    B = 3
    _, q, rp, ci, v, l, u, A = random_csr_qp_batch(B, n, m, density=max(dens, 1.5 / n), seed=3000 + t)
    shared = bool(t % 5 == 0)
    P = cases.sparse_spd(B, n, pd, seed=4000 + t, shared_pattern=shared)
    cp, ri, pv = cases.dense_to_csr(P)
    if shared: cp, ri = cp[0], ri[0]
    mode = t % 4
    outs = []
    for Parg in (P, (cp, ri, pv)):
        s = make_gpu(n, m, B)
        if mode == 0:
            s.settings.max_iter, s.settings.check_termination = 60, 0
        elif mode == 2:
            s.settings.adaptive_rho, s.settings.adaptive_rho_interval = 1, 25
        elif mode == 3:  # src/sqp.cpp:15-23
            s.settings.warm_start, s.settings.check_termination, s.settings.eps_abs, s.settings.eps_rel = 1, 10, 1e-4, 1e-4
            s.settings.max_iter, s.settings.adaptive_rho, s.settings.adaptive_rho_interval, s.settings.alpha = 100, 1, 50, 1.6
        if t % 2:
            s.setup_solve_csr(Parg, q, rp, ci, v, l, u)
        else:  # the stateful calls: set-up, solve, then an update with scaled matrices and a warm-started solve in one launch
            s.setup_csr(Parg, q, rp, ci, v, l, u)
            s.solve_csr(Parg, q, rp, ci, v, l, u)
            P2 = (Parg[0], Parg[1], 1.25 * Parg[2]) if isinstance(Parg, tuple) else 1.25 * Parg
            s.update_solve_csr(P2, 0.5 * q, rp, ci, v, l, u)
        outs.append((s.solution(), s.kernel_name(), s.settings))
        s.close()
    (x0, y0, z0, i0), k0, st = outs[0]
    (x1, y1, z1, i1), k1, _ = outs[1]
    kern[k1] = kern.get(k1, 0) + 1
    tot += B
    same = np.array_equal(x0, x1) and np.array_equal(y0, y1) and np.array_equal(z0, z1) and (i0.iter == i1.iter).all() and \
        (i0.status == i1.status).all() and (i0.rho_updates == i1.rho_updates).all() and np.array_equal(i0.res_prim, i1.res_prim) and \
        np.array_equal(i0.res_dual, i1.res_dual)
    if not same:
        differ += 1
        notes.append(("sparse != dense", n, m, pd, mode, k0, k1))
    if t % 2:  # the fused call against the oracle
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(st))
        ok = (i1.status == io["status"]).all() and (i1.iter == io["iter"]).all() and cases.relerr(x1, xo) < 1e-6 and cases.relerr(y1, yo) < 1e-6
        if not ok:
            orc_bad += 1
            notes.append(("oracle", n, m, pd, mode, k1, list(i1.iter), list(io["iter"])))
print("kernels (sparse-P calls):", sorted(kern.items()))
print("shapes 48, QPs %d: sparse-P call differing from the dense-P call in any bit: %d shapes; fused calls off the oracle (status / iterations / 1e-6): %d shapes" % (tot, differ, orc_bad))
for r in notes[:12]: print(r)
