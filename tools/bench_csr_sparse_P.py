"""Config 5's shape with P sparse as well (the legacy sparse class's Eigen::SparseMatrix P): the block-row kernel's sparse-P
instantiations (compressed columns read in place) against the same matrix handed over dense, device resident.  Prints kernel ms (HIP
events), the bytes that cross the boundary per QP, and whether the two results are bit-identical.
Usage: python tools/bench_csr_sparse_P.py [--batch 8192] [--pdensity 0.03] [--mode fixed|default]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_csr  # noqa: E402
from sqp_solver_amd import QPSolverBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--m", type=int, default=400)
    ap.add_argument("--pdensity", type=float, default=0.03)
    ap.add_argument("--mode", default="fixed")
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, m, B = a.n, a.m, a.batch
    _, q, rp, ci, v, l, u, A, nnz = bench_csr.make(B, n, m, 0.05, 20250233, dev)
    P, (colptr, rowind, pval), pnnz = bench_csr.make_sparse_P(B, n, a.pdensity, 7, dev)
    out = {}
    for name, Parg in (("dense P", P), ("sparse P", (colptr, rowind, pval))):
        s = QPSolverBatch(n, m, B)
        if a.mode == "fixed":
            s.settings.max_iter, s.settings.check_termination = 200, 0
        s.set_stream(torch.cuda.current_stream().cuda_stream)
        s.setup_solve_csr(Parg, q, rp, ci, v, l, u, colmajor=True)
        torch.cuda.synchronize()
        s.enable_timing(True)
        for _ in range(a.steps):
            s.setup_solve_csr(Parg, q, rp, ci, v, l, u, colmajor=True)
        torch.cuda.synchronize()
        ms = s.collect_kernel_ms()
        x, y, z, info = s.solution()
        out[name] = (x, y, info)
        print("%-9s kernel %-12s %7.3f ms per batch of %d (min %.3f)  mean iterations %.1f" % (name, s.kernel_name(), float(np.mean(ms)), B, float(np.min(ms)), info.iter.mean()))
    same = np.array_equal(out["dense P"][0], out["sparse P"][0]) and np.array_equal(out["dense P"][1], out["sparse P"][1]) and \
        (out["dense P"][2].iter == out["sparse P"][2].iter).all()
    dense_bytes = 8 * (n * n + n + 2 * m) + 12 * nnz + 4 * (m + 1) + 8 * (n + m) + 40
    sparse_bytes = 12 * pnnz + 4 * (n + 1) + 8 * (n + 2 * m) + 12 * nnz + 4 * (m + 1) + 8 * (n + m) + 40
    print("bit-identical x, y, iterations: %s;  nnz(P) %.0f of %d;  algorithmic bytes per QP %.0f (dense P) -> %.0f (sparse P)" % (same, pnnz, n * n, dense_bytes, sparse_bytes))


if __name__ == "__main__":
    main()
