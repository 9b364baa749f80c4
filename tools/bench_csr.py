"""BASELINE config 5 timing: batch x (n=200, m=400) QPs with a 5 %-dense CSR constraint matrix, device resident.
Usage: python tools/bench_csr.py [--batch 8192] [--iters 200] [--steps 3]"""
import argparse
import os
import json
import time

import numpy as np
import torch

from sqp_solver_amd import QPSolverBatch


def make(batch, n, m, density, seed, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    f64 = torch.float64
    P = torch.empty((batch, n, n), dtype=f64, device=dev)
    A = torch.empty((batch, m, n), dtype=f64, device=dev)
    eye = torch.eye(n, dtype=f64, device=dev)
    for s in range(0, batch, 1024):
        e = min(batch, s + 1024)
        G = torch.randn((e - s, n, n), generator=g, dtype=f64, device=dev)
        Pc = G @ G.transpose(1, 2) / n + 0.1 * eye
        P[s:e] = 0.5 * (Pc + Pc.transpose(1, 2))
        if os.environ.get("SQPH_BENCH_REGULAR_PATTERN"):
            # what-if measurement only (balanced sparse phases): every row has exactly round(density n) entries and every column the
            # same number — row i holds columns (7 i + k n / per_row) mod n
            per_row = max(1, int(round(density * n)))
            ii = torch.arange(m, device=dev).view(m, 1)
            kk = torch.arange(per_row, device=dev).view(1, per_row)
            cols = (7 * ii + kk * (n // per_row)) % n
            mask = torch.zeros((e - s, m, n), dtype=torch.bool, device=dev)
            mask[:, ii.expand(m, per_row), cols] = True
        else:
            mask = torch.rand((e - s, m, n), generator=g, device=dev) < density
            forced = torch.randint(0, n, (e - s, m, 1), generator=g, device=dev)
            mask.scatter_(2, forced, True)
        A[s:e] = torch.randn((e - s, m, n), generator=g, dtype=f64, device=dev) * mask
    q = torch.randn((batch, n), generator=g, dtype=f64, device=dev)
    x0 = torch.randn((batch, n), generator=g, dtype=f64, device=dev)
    c = torch.einsum("bij,bj->bi", A, x0)
    l = c - torch.rand((batch, m), generator=g, dtype=f64, device=dev)
    u = c + torch.rand((batch, m), generator=g, dtype=f64, device=dev)
    nzmask = A != 0
    counts = nzmask.sum(dim=2)
    rowptr = torch.zeros((batch, m + 1), dtype=torch.int32, device=dev)
    rowptr[:, 1:] = counts.cumsum(dim=1).to(torch.int32)
    nnz = rowptr[:, -1].to(torch.int64)
    nnz_max = int(nnz.max())
    idx = nzmask.nonzero()  # sorted (b, i, j): CSR order inside each QP
    start = torch.zeros(batch, dtype=torch.int64, device=dev)
    start[1:] = nnz.cumsum(0)[:-1]
    pos = torch.arange(idx.shape[0], device=dev) - start[idx[:, 0]]
    colind = torch.zeros((batch, nnz_max), dtype=torch.int32, device=dev)
    val = torch.zeros((batch, nnz_max), dtype=f64, device=dev)
    colind[idx[:, 0], pos] = idx[:, 2].to(torch.int32)
    val[idx[:, 0], pos] = A[idx[:, 0], idx[:, 1], idx[:, 2]]
    return P, q, rowptr, colind, val, l, u, A, float(nnz.double().mean())


def make_sparse_P(batch, n, density, seed, dev):
    """Symmetric, strictly diagonally dominant P with about `density` of the off-diagonal entries present (one pattern per QP): the dense
    matrices [batch, n, n] and their compressed columns (colptr [batch, n + 1], rowind / val [batch, nnz_max]; sqph_csc_P), mean nnz."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    mask = torch.triu(torch.rand((batch, n, n), generator=g, device=dev) < density, 1)
    U = torch.randn((batch, n, n), generator=g, dtype=torch.float64, device=dev) * mask
    P = U + U.transpose(1, 2)
    P += torch.diag_embed(P.abs().sum(2) + 0.5 + torch.rand((batch, n), generator=g, dtype=torch.float64, device=dev))
    nz = P != 0
    colptr = torch.zeros((batch, n + 1), dtype=torch.int32, device=dev)
    colptr[:, 1:] = nz.sum(2).cumsum(1).to(torch.int32)  # symmetric: row counts == column counts
    pn = colptr[:, -1].to(torch.int64)
    idx = nz.nonzero()
    start = torch.zeros(batch, dtype=torch.int64, device=dev)
    start[1:] = pn.cumsum(0)[:-1]
    pos = torch.arange(idx.shape[0], device=dev) - start[idx[:, 0]]
    pmax = int(pn.max())
    rowind = torch.zeros((batch, pmax), dtype=torch.int32, device=dev)
    pval = torch.zeros((batch, pmax), dtype=torch.float64, device=dev)
    rowind[idx[:, 0], pos] = idx[:, 2].to(torch.int32)
    pval[idx[:, 0], pos] = P[idx[:, 0], idx[:, 1], idx[:, 2]]
    return P, (colptr, rowind, pval), float(pn.double().mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--m", type=int, default=400)
    ap.add_argument("--density", type=float, default=0.05)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", type=int, default=4, help="QPs compared with the CPU oracle")
    ap.add_argument("--cpu-sample", type=int, default=0, help="time the CPU oracle (all host cores) on this many QPs")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    P, q, rp, ci, v, l, u, A, nnz = make(a.batch, a.n, a.m, a.density, 20250233, dev)
    s = QPSolverBatch(a.n, a.m, a.batch)
    s.settings.max_iter = a.iters
    s.settings.check_termination = 0
    s.setup_solve_csr(P, q, rp, ci, v, l, u)
    torch.cuda.synchronize()
    s.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        s.setup_solve_csr(P, q, rp, ci, v, l, u)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps
    kms = s.collect_kernel_ms()
    out = {"workload": "config5 batch %d n=%d m=%d CSR density %.2f (%.0f nnz/QP), %d fixed iters" % (a.batch, a.n, a.m, a.density, nnz, a.iters),
           "kernel": s.kernel_name(), "ms_per_step_wall": wall * 1e3, "kernel_ms": float(np.mean(kms)),
           "qp_per_s": a.batch / wall, "admm_iter_per_s": a.batch * a.iters / wall}
    if a.check:
        import oracle

        k = a.check
        x, y, z, info = s.solution()
        h = lambda t: t[:k].cpu().numpy()  # noqa: E731
        st = oracle.default_settings(max_iter=a.iters, check_termination=0)
        xo, yo, zo, io = oracle.solve_batch(h(P), h(q), h(A), h(l), h(u), st)
        rel = lambda p, r: float(np.max(np.abs(p - r).max(axis=1) / np.abs(r).max(axis=1)))  # noqa: E731
        out["parity_max_rel_err_x"], out["parity_max_rel_err_y"] = rel(x[:k], xo), rel(y[:k], yo)
    if a.cpu_sample:
        import os

        import oracle

        k = min(a.cpu_sample, a.batch)
        h = lambda t: t[:k].cpu().numpy()  # noqa: E731
        st = oracle.default_settings(max_iter=a.iters, check_termination=0)
        args = [h(P), h(q), h(A), h(l), h(u)]
        t0 = time.perf_counter()
        oracle.solve_batch(*args, st)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / dt, "unit": "QP/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "%d QPs, oracle/qp_oracle.c (dense (n+m)^2 KKT LDL'), OpenMP over QPs, %.1f s" % (k, dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
