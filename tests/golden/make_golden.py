"""Regenerates tests/golden/*.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`). Inputs come from sqp_solver_amd.problems.random_qp_batch
with fixed seeds; expected outputs are the oracle's x, y, z, status, iter, residuals."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from golden_io import SETTING_KEYS  # noqa: E402
from sqp_solver_amd.problems import SIMPLE_QP, random_qp_batch  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, n, m, batch, seed, settings overrides
    ("c1_simple_qp_default", 2, 3, 1, None, dict(max_iter=1000)),
    ("c1_simple_qp_sqp_settings", 2, 3, 1, None, dict(warm_start=1, check_termination=10, eps_abs=1e-4, eps_rel=1e-4, max_iter=100, adaptive_rho=1, adaptive_rho_interval=50, alpha=1.6)),
    ("c2_n20_m40_fixed200", 20, 40, 16, 20250230, dict(max_iter=200, check_termination=0)),
    ("c2_n20_m40_default", 20, 40, 16, 20250230, dict()),
    ("c3_n50_m100_fixed200", 50, 100, 8, 20250231, dict(max_iter=200, check_termination=0)),
    ("c3_n50_m100_default", 50, 100, 8, 20250231, dict()),
    ("c3_n50_m100_adaptive", 50, 100, 8, 20250231, dict(adaptive_rho=1, rho=0.002)),
    ("c2_n20_m40_sqp_settings", 20, 40, 16, 20250230, dict(warm_start=1, check_termination=10, eps_abs=1e-4, eps_rel=1e-4, max_iter=100, adaptive_rho=1, adaptive_rho_interval=50, alpha=1.6, rho=0.01)),
]


def csr_case():
    """BASELINE config 5 in small: a CSR constraint matrix beyond the dense register-tiled shapes (native sparse kernel)"""
    from sqp_solver_amd.problems import random_csr_qp_batch

    n, m, B = 80, 160, 3
    P, q, rp, ci, v, l, u, A = random_csr_qp_batch(B, n, m, density=0.08, seed=20250233)
    for name, over in (("c5_csr_n80_m160_fixed100", dict(max_iter=100, check_termination=0)), ("c5_csr_n80_m160_default", dict())):
        st = oracle.default_settings(**over)
        x, y, z, info = oracle.solve_batch(P, q, A, l, u, st, nthreads=1)
        d = dict(n=n, m=m, P=P, q=q, A=A, l=l, u=u, x=x, y=y, z=z, status=info["status"], iter=info["iter"],
                 rho_updates=info["rho_updates"], res_prim=info["res_prim"], res_dual=info["res_dual"],
                 csr_rowptr=rp, csr_colind=ci, csr_val=v)
        for k in SETTING_KEYS:
            d["set_" + k] = getattr(st, k)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "status", np.bincount(info["status"], minlength=2), "iters", info["iter"].min(), info["iter"].max())


def tiny_cases():
    """BASELINE config 4's subproblem shapes (n = 2, m = 3 and n = 3, m = 3; one QP per lane on the GPU) under the SQP driver's QP
    settings (src/sqp.cpp:15-23) and with a fixed iteration count.  Only QPs with a stable reference answer are kept: the double
    oracle and its x87 extended-precision instance agree on status / iterations / rho updates and to 1e-9 on x (adaptive rho on
    QPs this small can hinge on a residual at rounding level)."""
    sqp = dict(warm_start=1, check_termination=10, eps_abs=1e-4, eps_rel=1e-4, max_iter=100, adaptive_rho=1, adaptive_rho_interval=50, alpha=1.6)
    ld = np.longdouble
    for name, n, m, B, seed, over in (("c4_n2_m3_sqp_settings", 2, 3, 96, 20250232, sqp), ("c4_n3_m3_sqp_settings", 3, 3, 96, 20250234, sqp),
                                      ("c4_n2_m3_fixed100", 2, 3, 64, 20250232, dict(max_iter=100, check_termination=0))):
        P, q, A, l, u = random_qp_batch(B, n, m, seed=seed)
        st = oracle.default_settings(**over)
        x, y, z, info = oracle.solve_batch(P, q, A, l, u, st, nthreads=1)
        x80, y80, z80, i80 = oracle.solve_batch(P.astype(ld), q.astype(ld), A.astype(ld), l.astype(ld), u.astype(ld), st, nthreads=1, dtype=ld)
        ex = np.max(np.abs(x - x80.astype(np.float64)), axis=1) / np.maximum(np.max(np.abs(x), axis=1), 1e-300)
        ey = np.max(np.abs(y - y80.astype(np.float64)), axis=1) / np.maximum(np.max(np.abs(y), axis=1), 1.0)
        keep = (info["status"] == i80["status"]) & (info["iter"] == i80["iter"]) & (info["rho_updates"] == i80["rho_updates"]) & (ex < 1e-9) & (ey < 1e-9)
        P, q, A, l, u, x, y, z, info = P[keep], q[keep], A[keep], l[keep], u[keep], x[keep], y[keep], z[keep], info[keep]
        d = dict(n=n, m=m, P=P, q=q, A=A, l=l, u=u, x=x, y=y, z=z, status=info["status"], iter=info["iter"],
                 rho_updates=info["rho_updates"], res_prim=info["res_prim"], res_dual=info["res_dual"])
        for k in SETTING_KEYS:
            d["set_" + k] = getattr(st, k)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "kept", int(keep.sum()), "of", B, "status", np.bincount(info["status"], minlength=2), "iters", info["iter"].min(), info["iter"].max())


def main():
    csr_case()
    tiny_cases()
    for name, n, m, B, seed, over in CASES:
        if seed is None:
            S = SIMPLE_QP
            P, q, A, l, u = (S[k][None] for k in ("P", "q", "A", "l", "u"))
        else:
            P, q, A, l, u = random_qp_batch(B, n, m, seed=seed)
        st = oracle.default_settings(**over)
        x, y, z, info = oracle.solve_batch(P, q, A, l, u, st, nthreads=1)
        d = dict(n=n, m=m, P=P, q=q, A=A, l=l, u=u, x=x, y=y, z=z, status=info["status"], iter=info["iter"],
                 rho_updates=info["rho_updates"], res_prim=info["res_prim"], res_dual=info["res_dual"])
        for k in SETTING_KEYS:
            d["set_" + k] = getattr(st, k)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "status", np.bincount(info["status"], minlength=2), "iters", info["iter"].min(), info["iter"].max(), "rho_updates", info["rho_updates"].max())


if __name__ == "__main__":
    main()
