"""SURVEY section 8 row f4 measured, not argued: what would fp32 do to the register-tiled kernels' results?

Three candidates against two yard-sticks, at the BASELINE dense shapes (20,40) and (50,100), fixed iteration count:
  storage   the fp64 kernel with its B / W' tiles rounded through fp32 once (= fp32 tile STORAGE, fp64 accumulation: the variant
            that would free ~77 VGPRs at C3) — SQPH_TILE_QUANT in admm_wg_kernel.h
  schur32   the Schur-ordered iteration entirely in fp32 (admm_generic_kernel<float, float>: same formulas as the tiled kernels)
  ref32     the float instantiation of the reference path itself (oracle, dtype float32: (n+m)^2 pivoted LDL', src/qp.cpp:385-386)
  truth     the fp64 oracle on the same (float-representable) inputs
Errors are max over the batch of the per-QP relative infinity-norm error of x and y.  The row's acceptance rule:
ship fp32 tile storage iff its error against the float reference stays within 4x the float reference's own error against fp64.

Backends: `--backend sim` (the host SIMT emulator executing the product's kernel sources: IEEE-identical arithmetic, any machine)
and `--backend gpu` (an experiment build with -DSQPH_F32_TILE_STORAGE through the C-ABI on the MI355X; `storage` only — the library
has no fp32-arithmetic kernel for these shapes, which is the point).  Writes one JSON record per shape."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel(a, b, floor=1e-300):
    import numpy as np

    den = np.maximum(np.max(np.abs(b), axis=1), floor)
    return float(np.max(np.max(np.abs(a - b), axis=1) / den))


def measure(backend, n, m, batch, iters, seed=11):
    import numpy as np

    import oracle
    from sqp_solver_amd.problems import random_qp_batch

    P, q, A, l, u = random_qp_batch(batch, n, m, seed=seed, dtype=np.float32)  # float-representable inputs for every candidate
    st64 = oracle.default_settings(max_iter=iters, check_termination=0)
    xt, yt, zt, _ = oracle.solve_batch(*[a.astype(np.float64) for a in (P, q, A, l, u)], st64)
    xr, yr, zr, _ = oracle.solve_batch(P, q, A, l, u, st64, dtype=np.float32)
    rec = {"n": n, "m": m, "batch": batch, "iters": iters, "backend": backend,
           "ref32_vs_truth": {"x": rel(xr, xt), "y": rel(yr, yt, 1e-3)}}

    def run(make):
        s = make()
        s.settings.max_iter, s.settings.check_termination = iters, 0
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        return x.astype(np.float64), y.astype(np.float64), s

    if backend == "sim":
        import simlib

        variant = simlib.WG
        simlib.lib().sim_set_tile_quant(0)
        x0, y0, _ = run(lambda: simlib.SimSolverBatch(n, m, batch, dtype=np.float32, variant=variant))
        simlib.lib().sim_set_tile_quant(1)
        try:
            xs, ys, _ = run(lambda: simlib.SimSolverBatch(n, m, batch, dtype=np.float32, variant=variant))
        finally:
            simlib.lib().sim_set_tile_quant(0)
        xg, yg, _ = run(lambda: simlib.SimSolverBatch(n, m, batch, dtype=np.float32, variant=simlib.GENERIC_F32_ARITH, nt=64))
        rec["schur32_vs_truth"] = {"x": rel(xg, xt), "y": rel(yg, yt, 1e-3)}
        rec["schur32_vs_ref32"] = {"x": rel(xg, xr), "y": rel(yg, yr, 1e-3)}
        rec["kernel"] = "wg"
    else:
        from sqp_solver_amd import QPSolverBatch

        xs, ys, s = run(lambda: QPSolverBatch(n, m, batch, dtype=np.float32, device=0))
        rec["kernel"] = s.kernel_name()
        x0 = y0 = None
    rec["storage_vs_truth"] = {"x": rel(xs, xt), "y": rel(ys, yt, 1e-3)}
    rec["storage_vs_ref32"] = {"x": rel(xs, xr), "y": rel(ys, yr, 1e-3)}
    if x0 is not None:
        rec["fp64_kernel_vs_truth"] = {"x": rel(x0, xt), "y": rel(y0, yt, 1e-3)}
    worst = max(rec["storage_vs_truth"]["x"] / max(rec["ref32_vs_truth"]["x"], 1e-300), rec["storage_vs_truth"]["y"] / max(rec["ref32_vs_truth"]["y"], 1e-300))
    rec["storage_error_over_ref32_error"] = worst
    rec["ship_fp32_tile_storage"] = bool(worst <= 4.0)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["sim", "gpu"], default="sim")
    ap.add_argument("--out", default="")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    recs = []
    for (n, m) in ((20, 40), (50, 100)):
        batch = args.batch or (64 if args.backend == "gpu" else 6)
        recs.append(measure(args.backend, n, m, batch, args.iters))
        print(json.dumps(recs[-1]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(recs, f, indent=1)


if __name__ == "__main__":
    main()
