"""GPU parity tests (run with -m gpu on an MI355X): the HIP kernels through the C-ABI
(libsqp_hip.so via sqp_solver_amd.QPSolverBatch) against the CPU oracle on identical inputs."""
import os
import sys

import numpy as np
import pytest

import cases
import oracle

pytestmark = pytest.mark.gpu


def make_gpu(n, m, batch, dtype=np.float64, legacy_cold_start=False, force_generic=False, keep_factor=False, f32_arith=False, **kw):
    from sqp_solver_amd import QPSolverBatch

    return QPSolverBatch(n, m, batch, dtype=dtype, device=0, legacy_cold_start=legacy_cold_start, force_generic=force_generic,
                         keep_factor=keep_factor, f32_arith=f32_arith)


def make_gpu_generic(n, m, batch, **kw):
    return make_gpu(n, m, batch, force_generic=True, **kw)


MAKERS = [make_gpu, make_gpu_generic]
IDS = ["auto", "generic"]


def test_native_library_is_what_runs():
    import ctypes

    from sqp_solver_amd import _capi

    L = _capi.load()
    assert isinstance(L, ctypes.CDLL) and L.sqph_version() == 1
    s = make_gpu(2, 3, 1)
    qp = cases.simple()
    s.setup_solve(*qp)
    assert s.kernel_name() != "none"


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
@pytest.mark.parametrize("case", cases.REFERENCE_CASES, ids=lambda f: f.__name__)
def test_reference_cases(case, make):
    case(make)


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
@pytest.mark.parametrize("n,m,batch,iters", [(2, 3, 7, 200), (20, 40, 128, 200), (50, 100, 64, 200), (13, 57, 16, 100), (56, 104, 8, 50), (64, 30, 8, 50),
                                             (60, 250, 3, 40), (120, 250, 3, 40), (60, 600, 2, 30)])  # the last three: beyond the register-tiled shapes (CU-wide dense-A kernel; m > 512: generic)
def test_parity_fixed_iters(n, m, batch, iters, make):
    cases.parity_fixed_iters(make, n, m, batch, iters=iters)


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
def test_parity_fixed_alpha_and_float(make):
    cases.parity_fixed_iters(make, 20, 40, 32, iters=100, alpha=1.6)
    cases.parity_fixed_iters(make, 20, 40, 32, iters=100, dtype=np.float32)


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
@pytest.mark.parametrize("kw", [dict(), dict(adaptive=True), dict(sqp_settings=True)], ids=["default", "adaptive", "sqp"])
@pytest.mark.parametrize("n,m,batch", [(20, 40, 96), (50, 100, 48)])
def test_parity_termination(n, m, batch, kw, make):
    cases.parity_termination(make, n, m, batch, **kw)


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
def test_state_paths(make):
    cases.warm_start_and_resolve(make)
    cases.set_state_warm_start(make)
    cases.uninitialized_and_numerical_issues(make)
    for (n, m) in ((2, 3), (8, 12), (20, 40), (50, 100), (56, 112), (60, 120), (64, 128), (100, 100), (112, 128), (100, 200)):  # every kernel family's factorisation, the MFMA set-up among them (4, 4 and 7 block columns)
        cases.failing_pivots(make, n=n, m=m, batch=5)
    cases.shared_matrices(make)
    cases.edge_shapes(make)


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
@pytest.mark.parametrize("n,m", [(2, 3), (8, 12), (20, 40), (50, 100), (60, 120), (100, 100), (100, 200)])
def test_fused_call_then_solve(n, m, make):
    """a fused setup_solve keeps no factor by default; the following solve() rebuilds it (every kernel family)"""
    cases.fused_then_solve(make, n=n, m=m, batch=5)
    cases.solve_with_other_P(make, n=n, m=m, batch=5)  # solve(qp) with another P than setup(qp) (src/qp.cpp:324,360)


@pytest.mark.parametrize("n,m", [(2, 3), (4, 6), (8, 12), (20, 40), (50, 100), (100, 200)])
def test_setup_solve_reuse(n, m):
    """sqph_setup_solve_reuse (the SQP second-order correction): bit-identical to a plain setup+solve, every kernel family"""
    cases.soc_factor_reuse(make_gpu, n=n, m=m, batch=5)


@pytest.mark.parametrize("n,m", [(2, 3), (4, 6), (8, 12), (50, 100)])
def test_setup_solve_reuse_after_failed_setup(n, m):
    """the SOC fast path refactors a QP whose set-up ended in NUMERICAL_ISSUES (reference: sqp.cpp:274 -> 221-229)"""
    cases.soc_reuse_after_failed_setup(make_gpu, n=n, m=m, batch=5)


def test_four_wave_shapes():
    """m <= 224, 56 < n <= 112: the 32 x 16 lane grid, eight wavefronts per QP (7 x 7 + 4 x 7 doubles of tiles per lane)"""
    s = make_gpu(100, 200, 2)
    s.setup_solve(*[a[:2] for a in cases.random_qp_batch(2, 100, 200, seed=3)])
    assert s.kernel_name().startswith("wg8_32x16_7x7"), s.kernel_name()
    cases.parity_fixed_iters(make_gpu, 100, 200, 8, iters=60)
    cases.parity_fixed_iters(make_gpu, 112, 208, 4, iters=40)
    cases.parity_fixed_iters(make_gpu, 112, 224, 3, iters=40)
    cases.failing_pivots(make_gpu, n=100, m=200, batch=3)
    cases.parity_termination(make_gpu, 90, 180, 6, adaptive=True)


def test_grid_selection():
    """the lane grid each shape region takes (first fit in SQPH_WG_SHAPES; the stacked operator where m leaves room for W')"""
    for (n, m, name) in ((20, 40, "wg1_8x8_5x3"), (24, 96, "wg2_16x8_8x4_w3"), (32, 128, "wg2_16x8_8x4_w3"), (50, 100, "wg2_16x8_7x7s"), (56, 112, "wg2_16x8_7x7_"),
                         (60, 120, "wg4_16x16_8x4"), (100, 30, "wg4_16x16_2x7_w2"), (100, 100, "wg4_16x16_8x7_w2"), (100, 200, "wg8_32x16_7x7"), (50, 200, "wg4_32x8_7x7"),
                         (50, 400, "wg8_64x8_7x7"), (130, 150, "cud_t7"), (250, 300, "cud_t8"), (256, 512, "cud_t8"), (257, 300, "generic"), (200, 513, "generic")):
        s = make_gpu(n, m, 2)
        s.settings.max_iter, s.settings.check_termination = 5, 0
        s.setup_solve(*[a[:2] for a in cases.random_qp_batch(2, n, m, seed=3)])
        assert s.kernel_name().startswith(name), (n, m, s.kernel_name())


def test_dense_shapes_beyond_the_register_tiled_kernels():
    """112 < n <= 224 (or m beyond the tiled shapes' rows), m <= 512: the CU-wide kernel in its dense-A mode (csr_dense.hip: W in the
    CU's registers, A streamed from global memory twice per iteration) — fixed iterations, termination with adaptive rho, the
    stateful call sequences at its limits; beyond (m > 512 or n > 224) the generic kernel"""
    from sqp_solver_amd.problems import random_qp_batch

    for (n, m, b, kern) in ((120, 260, 4, "cud_t4"), (200, 400, 3, "cud_t7"), (50, 500, 3, "cud_t4"), (224, 512, 2, "cud_t7"), (113, 1, 3, "cud_t4"),
                            (230, 100, 2, "cud_t8"), (250, 300, 3, "cud_t8"), (256, 512, 2, "cud_t8"), (260, 100, 2, "generic"), (100, 520, 2, "generic")):
        cases.parity_fixed_iters(make_gpu, n, m, b, iters=40)
        s = make_gpu(n, m, b)
        s.settings.max_iter, s.settings.check_termination = 5, 0
        s.setup_solve(*random_qp_batch(b, n, m, seed=1))
        assert s.kernel_name().startswith(kern), (n, m, s.kernel_name())
    cases.parity_termination(make_gpu, 150, 300, 4, adaptive=True)
    cases.parity_termination(make_gpu, 200, 400, 3, sqp_settings=True)
    cases.fused_then_solve(make_gpu, n=130, m=200, batch=3)
    cases.warm_start_and_resolve(make_gpu, n=130, m=200)
    log, kernels = cases.api_sequence_fuzz(make_gpu, 224, 512, 2, seed=77, steps=6)
    assert any(k.startswith("cud_t7") for k in kernels), kernels
    # 224 < n <= 256 (round 6: the tile edge 8 of the same kernel; before, these shapes fell to the generic kernel)
    cases.parity_termination(make_gpu, 250, 300, 4, adaptive=True)
    cases.parity_termination(make_gpu, 256, 512, 3, sqp_settings=True)
    log, kernels = cases.api_sequence_fuzz(make_gpu, 256, 512, 2, seed=78, steps=6)
    assert any(k.startswith("cud_t8") for k in kernels), kernels


def test_dense_250_300_is_off_the_generic_kernel():
    """The reference class is Eigen::Dynamic (include/solvers/qp.hpp:118-131): 64 x (n=250, m=300), 100 fixed iterations — 16.7 ms on the
    generic kernel in round 5 (the CPU's rate), now the CU-wide kernel at tile edge 8: a generous bound on the kernel time guards the
    route (measured 3.4 ms), the whole batch is compared with the oracle."""
    from sqp_solver_amd.problems import random_qp_batch

    n, m, B = 250, 300, 64
    P, q, A, l, u = random_qp_batch(B, n, m, seed=41)
    s = make_gpu(n, m, B)
    s.settings.max_iter, s.settings.check_termination = 100, 0
    s.setup_solve(P, q, A, l, u)
    s.enable_timing(True)
    for _ in range(3):
        s.setup_solve(P, q, A, l, u)
    ms = min(s.collect_kernel_ms()[-3:])
    assert s.kernel_name().startswith("cud_t8"), s.kernel_name()
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), nthreads=0)
    assert cases.relerr(x, xo) < cases.TOL_F64 and cases.relerr(y, yo) < cases.TOL_F64 and cases.relerr(z, zo) < cases.TOL_F64
    print("64 x (250, 300), 100 iterations: %.2f ms" % ms)
    assert ms < 6.0, ms


def test_dense_300_600_on_the_generic_kernel():
    """beyond n = 256 or m = 512 the generic kernel (matrices in global memory): 64 x (300, 600) against the oracle at a fixed iteration
    count and under the default termination with adaptive rho"""
    cases.parity_fixed_iters(make_gpu, 300, 600, 64, iters=50)
    cases.parity_termination(make_gpu, 300, 600, 16, adaptive=True)


def test_csr_reference_cases():
    """tests/qp_solver_sparse_test.cpp through the CSR entry points"""
    cases.csr_reference_cases(make_gpu)
    cases.csr_malformed(make_gpu)


def make_gpu_csr_expand(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    from sqp_solver_amd import QPSolverBatch

    return QPSolverBatch(n, m, batch, dtype=dtype, device=0, legacy_cold_start=legacy_cold_start, csr_expand=True)


@pytest.mark.parametrize("make", [make_gpu, make_gpu_csr_expand], ids=["native", "expand"])
@pytest.mark.parametrize("n,m,batch,density,shared", [(20, 40, 16, 0.2, False), (50, 100, 16, 0.1, False), (30, 45, 8, 0.3, True),
                                                       (100, 180, 6, 0.08, False), (70, 300, 5, 0.1, True), (200, 400, 4, 0.05, False),
                                                       (224, 512, 3, 0.03, False)])
def test_csr_parity(n, m, batch, density, shared, make):
    """BASELINE config 5 shapes (n=200, m=400, 5 % dense CSR A) and smaller ones against the oracle on the densified A;
    shapes beyond the dense register-tiled kernels take the native sparse kernel (admm_csr_kernel.h)"""
    cases.csr_parity(make, n, m, batch, density=density, shared_pattern=shared, iters=50)


def test_csr_falls_back_when_the_sparse_matrix_does_not_fit_lds():
    """n=200, m=400 at 12 % density (~9,600 nnz > the ~6,000 the CU's LDS holds next to the vectors): expanded to dense on the device,
    then the dense dispatch — which at this shape is the CU-wide kernel's dense-A mode (round 2: the generic kernel)"""
    from sqp_solver_amd.problems import random_csr_qp_batch

    n, m, B = 200, 400, 2
    P, q, rp, ci, v, l, u, A = random_csr_qp_batch(B, n, m, density=0.12, seed=2)
    s = make_gpu(n, m, B)
    s.settings.max_iter, s.settings.check_termination = 30, 0
    s.setup_solve_csr(P, q, rp, ci, v, l, u)
    assert s.kernel_name().startswith("cud_t7"), s.kernel_name()
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings))
    assert cases.relerr(x, xo) < cases.TOL_F64 and cases.relerr(y, yo) < cases.TOL_F64


def test_csr_checking_instantiation_is_not_slower_per_iteration():
    """a guard, not a benchmark: the block-row kernel's checking instantiation once kept its register-resident slices of A in scratch
    (7x the no-check kernel's time per iteration, found only by timing it); 100 iterations with four primal-only checks must stay
    within 1.5x of 100 unchecked iterations"""
    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_csr

    B, n, m = 1024, 200, 400
    P, q, rp, ci, v, l, u, A, nnz = bench_csr.make(B, n, m, 0.05, 5, torch.device("cuda:0"))
    ms = {}
    for name, ct in (("fixed", 0), ("checked", 25)):
        s = make_gpu(n, m, B)
        s.settings.max_iter = 100
        s.settings.check_termination = ct
        s.settings.eps_abs = s.settings.eps_rel = 1e-300  # (no check ever passes)
        s.setup_solve_csr(P, q, rp, ci, v, l, u, colmajor=True)
        s.enable_timing(True)
        for _ in range(3):
            s.setup_solve_csr(P, q, rp, ci, v, l, u, colmajor=True)
        ms[name] = float(np.median(s.collect_kernel_ms()[-3:]))
        assert s.kernel_name().startswith("csb_")
        s.close()
    assert ms["checked"] <= 1.5 * ms["fixed"], ms


def test_csr_edge_cases():
    cases.csr_edge_cases(make_gpu)


@pytest.mark.parametrize("n,m,density", [(200, 400, 0.05), (30, 45, 0.3), (100, 180, 0.08)])
def test_csr_update_solve(n, m, density):
    """sqph_update_solve_csr on the block-row kernel and on the expand + dense route"""
    cases.csr_update_solve(make_gpu, n=n, m=m, density=density)


@pytest.mark.parametrize("n,m,density,sparse_P", [(200, 400, 0.05, False), (200, 400, 0.05, True), (30, 45, 0.3, False), (100, 180, 0.08, False)])
def test_csr_setup_solve_reuse(n, m, density, sparse_P):
    """sqph_setup_solve_reuse_csr / _csr_sp (the SOC re-solve of a sparse subproblem, src/sqp.cpp:244-276, TODO :273): bit-identical to
    sqph_setup_solve_csr on the block-row kernel (nb13 / nb9) and on the expand + dense route ((30, 45): register-tiled kernel)"""
    cases.soc_factor_reuse_csr(make_gpu, n=n, m=m, batch=3, density=density, sparse_P=sparse_P)
    if (n, m) == (200, 400) and not sparse_P:
        # the fast path is taken, not just harmless: with one ADMM iteration per QP a call is its set-up, and the reuse call skips it
        from sqp_solver_amd.problems import random_csr_qp_batch

        B = 512
        P, q, rp, ci, v, l, u, A = random_csr_qp_batch(8, n, m, density=density, seed=3)
        rep = lambda a: np.concatenate([a] * (B // 8))  # noqa: E731
        args = [rep(a) for a in (P, q, rp, ci, v, l, u)]
        s = make_gpu(n, m, B, keep_factor=True)
        s.settings.max_iter, s.settings.check_termination = 1, 0
        s.setup_solve_csr(*args)
        s.enable_timing(True)
        for _ in range(3):
            s.setup_solve_csr(*args)
        for _ in range(3):
            s.setup_solve_reuse_csr(*args)
        ms = s.collect_kernel_ms()
        full, reuse = min(ms[-6:-3]), min(ms[-3:])
        print("setup_solve_csr %.3f ms, setup_solve_reuse_csr %.3f ms (512 x (200, 400), 1 iteration)" % (full, reuse))
        assert reuse < 0.6 * full, (full, reuse)


def test_csr_sparse_P():
    """sqph_*_csr_sp: P in compressed-column form, bit-identical to the dense-P twin on every CSR route (block-row kernel, expand +
    dense), the stateful calls, the reference's sparse test problem, malformed structures"""
    cases.csr_sparse_P(make_gpu)


def test_csr_native_kernel_is_used_and_handles_termination_paths():
    from sqp_solver_amd.problems import random_csr_qp_batch

    n, m, B = 200, 400, 6
    P, q, rp, ci, v, l, u, A = random_csr_qp_batch(B, n, m, density=0.05, seed=9)
    for kw in (dict(), dict(adaptive_rho=1), dict(adaptive_rho=1, adaptive_rho_interval=10, alpha=1.6, eps_abs=1e-5, eps_rel=1e-5)):
        s = make_gpu(n, m, B)
        for k, val in kw.items():
            setattr(s.settings, k, val)
        s.setup_solve_csr(P, q, rp, ci, v, l, u)
        assert s.kernel_name() in ("csb_nb13", "csb_nb14")
        x, y, z, info = s.solution()
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings))
        assert (info.status == io["status"]).all() and (info.iter == io["iter"]).all() and (info.rho_updates == io["rho_updates"]).all()
        assert cases.relerr(x, xo) < cases.TOL_F64 and cases.relerr(y, yo) < cases.TOL_F64
    # setup, then solve twice (warm), then update_qp with new values on the same pattern
    s = make_gpu(n, m, B)
    so = [oracle.QPSolver() for _ in range(B)]
    s.setup_csr(P, q, rp, ci, v, l, u)
    s.solve_csr(P, q, rp, ci, v, l, u)
    s.solve_csr(P, q + 0.1, rp, ci, v, l, u)
    v2 = v * 1.25
    s.update_qp_csr(P, q, rp, ci, v2, l, u)
    s.solve_csr(P, q, rp, ci, v2, l, u)
    x, y, z, info = s.solution()
    for b in range(B):
        o = so[b]
        o.setup(P[b], q[b], A[b], l[b], u[b])
        o.solve(P[b], q[b], A[b], l[b], u[b])
        o.solve(P[b], q[b] + 0.1, A[b], l[b], u[b])
        o.update_qp(P[b], q[b], 1.25 * A[b], l[b], u[b])
        o.solve(P[b], q[b], 1.25 * A[b], l[b], u[b])
        assert info.iter[b] == o.info.iter and info.status[b] == o.info.status
        assert cases.relerr(x[b], o.primal_solution()) < cases.TOL_F64 and cases.relerr(y[b], o.dual_solution()) < cases.TOL_F64


def test_golden_fixtures():
    import golden_io

    for name, g in golden_io.load_all():
        s = make_gpu(g["n"], g["m"], g["P"].shape[0])
        golden_io.apply_settings(s.settings, g)
        if "csr_rowptr" in g:
            s.setup_solve_csr(g["P"], g["q"], g["csr_rowptr"], g["csr_colind"], g["csr_val"], g["l"], g["u"])
            assert s.kernel_name().startswith(("csr_", "csb_")), s.kernel_name()
        else:
            s.setup_solve(g["P"], g["q"], g["A"], g["l"], g["u"])
        x, y, z, info = s.solution()
        assert cases.relerr(x, g["x"]) < cases.TOL_F64, name
        assert np.max(np.abs(y - g["y"])) <= cases.TOL_F64 * max(1.0, np.max(np.abs(g["y"]))), name
        assert (info.status == g["status"]).all() and (info.iter == g["iter"]).all(), name


def test_full_size_properties_c3():
    """BASELINE config 3 shard (8,192 x n=50,m=100), default termination: size-independent properties.
    (1) every SOLVED QP passes the reference's termination test recomputed in numpy;
    (2) a sample agrees with the oracle; (3) solving the reversed batch reverses the results bit-exactly
    (QPs are independent: no cross-QP coupling in the kernel)."""
    import torch

    from sqp_solver_amd.problems import random_qp_batch_torch

    B, n, m = 8192, 50, 100
    P, q, A_cm, l, u = random_qp_batch_torch(B, n, m, seed=123)
    s = make_gpu(n, m, B)
    s.setup_solve(P, q, A_cm, l, u, colmajor=True)
    x, y, z, info = s.solution()
    assert np.isin(info.status, [0, 1]).all()
    assert (info.status == 0).mean() > 0.95
    Ph, qh = P.cpu().numpy().transpose(0, 2, 1), q.cpu().numpy()
    Ah, lh, uh = A_cm.cpu().numpy().transpose(0, 2, 1), l.cpu().numpy(), u.cpu().numpy()
    rp, rd, ep, ed = cases.kkt_property(x, y, z, Ph, qh, Ah, lh, uh, 1e-3, 1e-3)
    ok = info.status == 0
    assert (rp[ok] <= ep[ok] * (1 + 1e-9)).all() and (rd[ok] <= ed[ok] * (1 + 1e-9) + 1e-12).all()
    assert ((z >= lh - 1e-12) & (z <= uh + 1e-12)).all()  # z is a projection onto [l,u]
    k = 96
    xo, yo, zo, io = oracle.solve_batch(Ph[:k], qh[:k], Ah[:k], lh[:k], uh[:k], oracle.default_settings())
    assert cases.relerr(x[:k], xo) < cases.TOL_F64 and cases.relerr(y[:k], yo) < cases.TOL_F64
    assert (info.status[:k] == io["status"]).all() and (info.iter[:k] == io["iter"]).all()
    rev = lambda t: torch.flip(t, dims=[0]).contiguous()  # noqa: E731
    s.setup_solve(rev(P), rev(q), rev(A_cm), rev(l), rev(u), colmajor=True)
    x2, y2, z2, info2 = s.solution()
    assert np.array_equal(x2[::-1], x) and np.array_equal(y2[::-1], y) and np.array_equal(info2.iter[::-1], info.iter)


def test_full_size_properties_c5():
    """BASELINE config 5 at full size (8,192 x n=200, m=400, 5 % dense CSR A), default termination, native sparse kernel:
    every SOLVED QP passes the reference's termination test recomputed with torch on the dense A the CSR arrays encode;
    z is a projection; a sample agrees with the oracle; the reversed batch reverses the results bit-exactly."""
    import sys

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_csr

    B, n, m = 8192, 200, 400
    dev = torch.device("cuda:0")
    P, q, rp, ci, v, l, u, A, nnz = bench_csr.make(B, n, m, 0.05, 99, dev)
    s = make_gpu(n, m, B)
    s.setup_solve_csr(P, q, rp, ci, v, l, u)
    assert s.kernel_name() in ("csb_nb13", "csb_nb14")
    x, y, z, info = s.solution()
    assert np.isin(info.status, [0, 1]).all() and (info.status == 0).mean() > 0.9
    xt, yt, zt = (torch.from_numpy(a).to(dev) for a in (x, y, z))
    Ax = torch.einsum("bij,bj->bi", A, xt)
    Px = torch.einsum("bij,bj->bi", P, xt)
    ATy = torch.einsum("bij,bi->bj", A, yt)
    nrm = lambda t: t.abs().amax(dim=1)  # noqa: E731
    r_p, r_d = nrm(Ax - zt), nrm(Px + q + ATy)
    e_p = 1e-3 + 1e-3 * torch.maximum(nrm(Ax), nrm(zt))
    e_d = 1e-3 + 1e-3 * torch.maximum(nrm(Px), torch.maximum(nrm(ATy), nrm(q)))
    ok = torch.from_numpy(info.status == 0).to(dev)
    assert bool((r_p[ok] <= e_p[ok] * (1 + 1e-9)).all()) and bool((r_d[ok] <= e_d[ok] * (1 + 1e-9) + 1e-12).all())
    assert bool(((zt >= l - 1e-12) & (zt <= u + 1e-12)).all())
    k = 8
    h = lambda t: t[:k].cpu().numpy()  # noqa: E731
    xo, yo, zo, io = oracle.solve_batch(h(P), h(q), h(A), h(l), h(u), oracle.default_settings())
    assert cases.relerr(x[:k], xo) < cases.TOL_F64 and cases.relerr(y[:k], yo) < cases.TOL_F64
    assert (info.status[:k] == io["status"]).all() and (info.iter[:k] == io["iter"]).all()
    rev = lambda t: torch.flip(t, dims=[0]).contiguous()  # noqa: E731
    s.setup_solve_csr(rev(P), rev(q), rev(rp), rev(ci), rev(v), rev(l), rev(u))
    x2, y2, z2, info2 = s.solution()
    assert np.array_equal(x2[::-1], x) and np.array_equal(y2[::-1], y) and np.array_equal(info2.iter[::-1], info.iter)


def test_small_shapes_take_the_lane_and_four_per_wave_kernels():
    """n <= 4 / m <= 6 (the SQP driver's subproblems): one QP per lane — four lanes per QP (the quad variant, m <= 4) up to 2,048
    QPs (the SQP driver's batch sizes); up to n = 12 / m = 24: four
    QPs per wavefront; large odd batches against the oracle"""
    from sqp_solver_amd.problems import random_qp_batch

    for (n, m, B, kern) in ((2, 3, 1999, "quad_2x3_exact"), (2, 3, 4099, "lane_2x3_exact"), (4, 6, 3001, "lane_4x6"), (3, 3, 1500, "quad_3x3_exact"),
                           (3, 3, 2100, "lane_3x3_exact"), (4, 4, 333, "quad_4x4"), (3, 2, 2500, "lane_4x4"), (4, 5, 700, "lane_4x6"), (8, 12, 2051, "g16_"), (12, 24, 1027, "g16_")):
        P, q, A, l, u = random_qp_batch(B, n, m, seed=31)
        s = make_gpu(n, m, B)
        s.setup_solve(P, q, A, l, u)
        assert s.kernel_name().startswith(kern), s.kernel_name()
        x, y, z, info = s.solution()
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, oracle.default_settings(), nthreads=0)
        assert (info.status == io["status"]).all() and (info.iter == io["iter"]).all()
        # tiny QPs can have every constraint inactive (y == 0 up to rounding): scale the dual error by max(1, |y|)
        assert cases.relerr(x, xo) < cases.TOL_F64
        assert (np.max(np.abs(y - yo), axis=1) <= cases.TOL_F64 * np.maximum(1.0, np.max(np.abs(yo), axis=1))).all()


def test_true_fp32_variant():
    """SQPH_FLAG_F32_ARITH on the shapes that have an fp32 kernel (one QP per lane): within TOL_F32 = 5e-3 of the
    reference's QPSolver<float> (float oracle), no further from the fp64 solution than 4x the float oracle is (floor 2e-3);
    status / iteration counts equal to the float oracle's; shapes without an fp32 kernel silently iterate in fp64."""
    mk = lambda n, m, b, **kw: make_gpu(n, m, b, dtype=np.float32, f32_arith=True)  # noqa: E731
    for (n, m) in ((2, 3), (4, 6), (3, 3)):
        ex, ey, ez = cases.parity_fixed_iters(mk, n, m, 2048, iters=150, dtype=np.float32, dual_floor=True, f32_floor=2e-3)
        assert ex < 2e-3 and ey < 2e-3, (ex, ey)
    s = mk(2, 3, 4)
    s.setup_solve(*cases.simple(4, dtype=np.float32))
    assert s.kernel_name() == "lane_2x3_exact_f32", s.kernel_name()
    from sqp_solver_amd.problems import random_qp_batch

    P, q, A, l, u = random_qp_batch(4096, 4, 6, seed=7, dtype=np.float32)
    s = mk(4, 6, 4096)
    s.setup_solve(P, q, A, l, u)
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), dtype=np.float32)
    assert ((info.status == io["status"]) & (info.iter == io["iter"])).mean() > 0.995  # fp32 stop tests sit on fp32-noisy residuals
    assert np.percentile(np.max(np.abs(x - xo), axis=1) / np.maximum(np.max(np.abs(xo), axis=1), 1e-30), 99) < cases.TOL_F32
    s = mk(60, 120, 8)  # no fp32 kernel for this shape: fp64 arithmetic behind the float interface, as without the flag
    s.settings.max_iter, s.settings.check_termination = 50, 0
    s.setup_solve(*random_qp_batch(8, 60, 120, seed=1, dtype=np.float32))
    assert not s.kernel_name().endswith("_f32")


def test_fp32_product_variant_of_the_register_tiled_kernels():
    """AN APPROXIMATE MODE, NOT A PARITY CLAIM (round 5: SURVEY section 8 row f4 asked for y within 2x of the float reference; this
    variant measures 2.7-4.6x and is documented as outside the parity contract in include/sqp_hip.h — the bars below PIN its measured
    behaviour so that it cannot silently get worse, they were set after measuring).
    SQPH_FLAG_F32_ARITH at the BASELINE dense shapes (reference src/qp.cpp:385-386 and
    tests/qp_solver_test.cpp:58-69): tiles, operands and partial sums of the two iteration stages in fp32 (wg_f32.hip), the
    factorisation, the iterates and the residual checks in fp64.  Stated accuracy, WITHOUT a floor (round 4; measured on the MI355X,
    tools/xp/f32_err.py: x 0.7-1.7x, z 0.5-1.8x, y 2.7-4.6x the float oracle's error against the fp64 solution): x and z no further
    from the fp64 solution than 2.5x the reference's QPSolver<float>, y no further than 6x — i.e. this variant is as accurate as the
    reference's float path on the primal side and ~4x less accurate on the dual.  Default termination against
    the FP64 oracle: status equal, iteration counts equal on >= 90 % of the batch, solutions of those within 1e-4."""
    from sqp_solver_amd.problems import random_qp_batch

    mk = lambda n, m, b, **kw: make_gpu(n, m, b, dtype=np.float32, f32_arith=True, keep_factor=kw.get("keep_factor", False))  # noqa: E731
    worst = {}
    for (n, m, b, kern) in ((20, 40, 512, "wg1_8x8_5x3_w3_f32"), (50, 100, 256, "wg2_16x8_7x7s_w2_f32"), (30, 60, 64, "wg2_16x8_7x7s_w2_f32"), (56, 112, 32, "wg2_16x8_7x7_w2_f32")):
        ex, ey, ez = cases.parity_fixed_iters(mk, n, m, b, iters=200, dtype=np.float32, f32_floor=0.0, f32_ratio=(2.5, 6.0, 2.5))
        assert ex < 2e-5 and ey < 1e-3, (n, m, ex, ey)
        worst[(n, m)] = (ex, ey)
        P, q, A, l, u = random_qp_batch(b, n, m, seed=3, dtype=np.float32)
        s = mk(n, m, b)
        s.setup_solve(P, q, A, l, u)
        assert s.kernel_name() == kern, s.kernel_name()
        x, y, z, info = s.solution()
        f64 = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        xo, yo, zo, io = oracle.solve_batch(f64(P), f64(q), f64(A), f64(l), f64(u), cases.oracle_settings(s.settings), nthreads=0)
        assert (info.status == io["status"]).mean() >= 0.98
        same = (info.iter == io["iter"]) & (info.status == io["status"])
        assert same.mean() >= 0.9, same.mean()
        assert cases.relerr(x[same], xo[same]) < 1e-4
        # solve() on the resident factor == the fused call, bit for bit, at a fixed iteration count (under termination the fused
        # call keeps A x by recurrence from the fp32 products, a solve() on retained iterates streams A: the stop test may differ)
        s.settings.max_iter, s.settings.check_termination = 60, 0
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        s2 = mk(n, m, b, keep_factor=True)
        s2.settings.max_iter, s2.settings.check_termination = 60, 0
        s2.setup(P, q, A, l, u)
        s2.solve(P, q, A, l, u)
        x2, y2, z2, info2 = s2.solution()
        assert np.array_equal(x2, x) and np.array_equal(y2, y) and np.array_equal(info2.iter, info.iter)
    print("fp32-product kernels, error against the fp64 solution (x, y):", worst)
    cases.ref_testSinglePrecisionFloat(mk)


def test_verbose_trace():
    """settings.verbose (reference print_status, src/qp.cpp:373-383): one record per termination check for the traced QP — the
    iteration, the objective, the residuals; the last one is the info record's; a verbose call composes with non-verbose ones
    on the same handle (the recording kernels are another kernel family: the factor is rebuilt across the switch)"""
    from sqp_solver_amd.problems import random_qp_batch

    for (n, m, B, tq) in ((2, 3, 5, 0), (20, 40, 7, 3), (50, 100, 4, 2)):
        P, q, A, l, u = (cases.simple(B) if n == 2 else random_qp_batch(B, n, m, seed=4))
        s = make_gpu(n, m, B)
        s.settings.verbose = 1
        s.set_trace_qp(tq)
        s.setup(P, q, A, l, u)
        s.solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        rec = s.trace()
        ct = s.settings.check_termination
        assert len(rec) == int(info.iter[tq]) // ct and len(rec) >= 1
        assert (rec[:, 0] == ct * np.arange(1, len(rec) + 1)).all()
        assert rec[-1, 2] == info.res_prim[tq] and rec[-1, 3] == info.res_dual[tq]
        obj = 0.5 * x[tq] @ P[tq] @ x[tq] + q[tq] @ x[tq]
        assert abs(rec[-1, 1] - obj) <= 1e-9 * max(1.0, abs(obj))
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings))
        assert (info.iter == io["iter"]).all() and cases.relerr(x, xo) < cases.TOL_F64
        # non-verbose solve() on the same handle afterwards (fast kernel, factor rebuilt), then verbose again
        s.settings.verbose = 0
        s.solve(P, q + 0.1, A, l, u)
        assert len(s.trace()) == 0
        x1, y1, z1, info1 = s.solution()
        s.settings.verbose = 1
        s.solve(P, q, A, l, u)
        assert len(s.trace()) >= 1
        o = oracle.QPSolver()
        o.setup(P[tq], q[tq], A[tq], l[tq], u[tq])
        o.solve(P[tq], q[tq], A[tq], l[tq], u[tq])
        o.solve(P[tq], q[tq] + 0.1, A[tq], l[tq], u[tq])
        assert info1.iter[tq] == o.info.iter and cases.relerr(x1[tq][None], o.primal_solution()[None]) < cases.TOL_F64
        o.solve(P[tq], q[tq], A[tq], l[tq], u[tq])
        assert s.info().iter[tq] == o.info.iter and cases.relerr(s.solution()[0][tq][None], o.primal_solution()[None]) < cases.TOL_F64


def test_lane_kernel_paths():
    """the one-QP-per-lane kernel through the C-ABI: fixed iterations, alpha, float interface, termination / adaptive / SQP
    settings (QPs without a stable reference answer excluded by parity_termination), state paths, fused-then-solve"""
    for (n, m, B) in ((2, 3, 257), (1, 1, 65), (2, 1, 64), (3, 3, 130), (4, 6, 300)):
        cases.parity_fixed_iters(make_gpu, n, m, B, iters=150, dual_floor=True)
    cases.parity_fixed_iters(make_gpu, 2, 3, 64, iters=100, alpha=1.6, dual_floor=True)  # (a tiny QP may have every constraint inactive: y = 0)
    cases.parity_fixed_iters(make_gpu, 4, 6, 64, iters=100, dtype=np.float32)
    for kw in (dict(), dict(adaptive=True), dict(sqp_settings=True)):
        # residual norms / rho estimate are compared on the QPs whose reference diagnostics are themselves reproducible
        cases.parity_termination(make_gpu, 4, 6, 200, diagnostics=True if not kw else "stable", **kw)
        cases.parity_termination(make_gpu, 2, 3, 200, diagnostics=True if not kw else "stable", **kw)
    cases.warm_start_and_resolve(make_gpu, n=4, m=6)
    cases.set_state_warm_start(make_gpu, n=3, m=5)
    cases.shared_matrices(make_gpu, n=4, m=6)
    cases.fused_then_solve(make_gpu, n=4, m=5, batch=70, adaptive=False)  # (adaptive rho on QPs this small: no stable reference answer)
    cases.fused_then_solve(make_gpu, n=4, m=5, batch=3)
    s = make_gpu(2, 3, 4)
    s.setup_solve(*cases.simple(4))
    assert s.kernel_name() == "quad_2x3_exact"  # (small batch: four lanes per QP)
    # the same paths on batches beyond the quad variant's limit (one QP per lane)
    cases.parity_fixed_iters(make_gpu, 2, 3, 2100, iters=100, dual_floor=True)
    cases.parity_termination(make_gpu, 2, 3, 2100, sqp_settings=True, diagnostics="stable")
    s = make_gpu(2, 3, 2100)
    s.setup_solve(*cases.random_qp_batch(2100, 2, 3, seed=2))
    assert s.kernel_name() == "lane_2x3_exact"


def test_full_size_fixed_iters_c2():
    """BASELINE config 2: 4,096 x (n=20, m=40), 200 ADMM iterations, whole batch against the oracle."""
    from sqp_solver_amd.problems import random_qp_batch

    B, n, m = 4096, 20, 40
    P, q, A, l, u = random_qp_batch(B, n, m, seed=20250230)
    s = make_gpu(n, m, B)
    s.settings.max_iter = 200
    s.settings.check_termination = 0
    s.setup_solve(P, q, A, l, u)
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), nthreads=0)
    assert cases.relerr(x, xo) < cases.TOL_F64 and cases.relerr(y, yo) < cases.TOL_F64
    assert (info.iter == 201).all()


def test_full_size_one_launch_c3_65536():
    """The headline workload itself (BASELINE configs[2]: 65,536 x (n=50, m=100), 200 fixed iterations, ONE launch — 4.1 GB of inputs,
    1.9 GB of state, per-QP offsets up to 65,535 x 5,000 doubles): (1) the reversed batch reverses the results bit for bit (the
    last QP of one launch is the first of the other: any 32-bit offset overflow or cross-QP coupling breaks this); (2) a 256-QP
    sample drawn from the LAST 4,096 QPs of the launch (the largest offsets) agrees with the oracle; (3) every QP ran 200 iterations."""
    import torch

    from sqp_solver_amd.problems import random_qp_batch_torch

    B, n, m = 65536, 50, 100
    dev = torch.device("cuda:0")
    P, q, A_cm, l, u = random_qp_batch_torch(B, n, m, seed=20250228 + 3, dtype=torch.float64, device=dev)
    s = make_gpu(n, m, B)
    s.settings.max_iter = 200
    s.settings.check_termination = 0
    s.setup_solve(P, q, A_cm, l, u, colmajor=True)
    assert s.kernel_name().startswith("wg2_16x8"), s.kernel_name()
    x, y, z, info = s.solution()
    assert (info.iter == 201).all() and (info.status == cases.MAX_ITER_EXCEEDED).all()
    assert np.isfinite(x).all() and np.isfinite(y).all()
    idx = np.sort(np.random.default_rng(7).choice(np.arange(B - 4096, B), size=256, replace=False))
    ti = torch.from_numpy(idx).to(dev)
    h = lambda t: t[ti].cpu().numpy()  # noqa: E731
    xo, yo, zo, io = oracle.solve_batch(h(P).transpose(0, 2, 1), h(q), h(A_cm).transpose(0, 2, 1), h(l), h(u),
                                        cases.oracle_settings(s.settings), nthreads=0)
    assert cases.relerr(x[idx], xo) < cases.TOL_F64 and cases.relerr(y[idx], yo) < cases.TOL_F64 and cases.relerr(z[idx], zo) < cases.TOL_F64
    assert (info.iter[idx] == io["iter"]).all() and (info.status[idx] == io["status"]).all()
    rev = lambda t: torch.flip(t, dims=[0]).contiguous()  # noqa: E731
    Pr, qr, Ar, lr, ur = rev(P), rev(q), rev(A_cm), rev(l), rev(u)
    del P, A_cm
    s.setup_solve(Pr, qr, Ar, lr, ur, colmajor=True)
    x2, y2, z2, info2 = s.solution()
    assert np.array_equal(x2[::-1], x) and np.array_equal(y2[::-1], y) and np.array_equal(z2[::-1], z)
    assert np.array_equal(info2.iter[::-1], info.iter)
    s.close()


def test_api_misuse_errors():
    from sqp_solver_amd import QPSolverBatch, SqphError

    s = QPSolverBatch(3, 4, 2)
    P, q, A, l, u = __import__("sqp_solver_amd.problems", fromlist=["x"]).random_qp_batch(3, 3, 4)
    with pytest.raises(ValueError):
        s.setup(P, q, A, l, u)  # batch 3 > capacity 2
    s.settings.alpha = 2.5
    with pytest.raises(SqphError):
        s.setup(P[:2], q[:2], A[:2], l[:2], u[:2])
    with pytest.raises(SqphError):
        QPSolverBatch(0, 1, 1)


def _stress_log(rec):
    import json

    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "stress_metrics.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


@pytest.mark.parametrize("make", MAKERS, ids=IDS)
@pytest.mark.parametrize("kind", cases.STRESS_KINDS)
@pytest.mark.parametrize("n,m,batch", [(20, 40, 32), (50, 100, 32)])
def test_stress_parity(n, m, batch, kind, make):
    """rho at its clamps, equality-heavy and ill-conditioned batches against the oracle (and its x87 yard-stick)"""
    cases.stress_parity(make, n, m, batch, kind, log=_stress_log)


def test_stress_parity_sparse_shape():
    """the same stress kinds on the config-5 shape through the native sparse kernel"""
    from sqp_solver_amd.problems import random_csr_qp_batch

    n, m, B = 200, 400, 4
    for kind in ("rho_low", "rho_high", "half_eq"):
        P, q, rp, ci, v, l, u, A = random_csr_qp_batch(B, n, m, density=0.05, seed=21)
        if kind == "half_eq":
            # equalities on every other row, feasible by construction: both bounds at the row's midpoint
            fin = np.isfinite(l) & np.isfinite(u) & (np.abs(l) < 1e19)
            c = np.where(fin, 0.5 * (l + u), 0.0)
            eq = fin & (np.arange(m) % 2 == 0)[None, :]
            l, u = np.where(eq, c, l), np.where(eq, c, u)
        s = make_gpu(n, m, B)
        cases.stress_settings(s.settings, kind, 100)
        s.setup_solve_csr(P, q, rp, ci, v, l, u)
        assert s.kernel_name() in ("csb_nb13", "csb_nb14")
        x, y, z, info = s.solution()
        ost = cases.oracle_settings(s.settings)
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, ost)
        ld = np.longdouble
        xl, yl, _, _ = oracle.solve_batch(P.astype(ld), q.astype(ld), A.astype(ld), l.astype(ld), u.astype(ld), ost, dtype=ld)
        noise = max(cases.relerr(xo, xl.astype(np.float64)), cases.relerr(yo, yl.astype(np.float64)))
        tol = max(cases.TOL_F64, 4 * noise)
        rec = dict(kind=kind + "_csr", n=n, m=m, batch=B, ex=cases.relerr(x, xo), ey=cases.relerr(y, yo), noise=noise, tol=tol)
        _stress_log(rec)
        assert rec["ex"] < tol and rec["ey"] < tol, rec
        assert (info.rho_updates == io["rho_updates"]).all() and (info.iter == io["iter"]).all()


def test_hip_graph_capture_of_the_fused_call():
    """Device-memspace calls only enqueue work on the handle's stream (no allocation, no synchronisation after the first call of a
    shape), so a fused setup+solve can be captured into a hipGraph and replayed on updated inputs — the launch-bound MPC pattern."""
    import torch

    from sqp_solver_amd.problems import random_qp_batch_torch

    n, m, B = 50, 100, 128
    P, q, A, l, u = random_qp_batch_torch(B, n, m, seed=3, device="cuda:0")
    s = make_gpu(n, m, B)
    s.settings.max_iter = 60
    s.settings.check_termination = 0
    st = torch.cuda.Stream()
    s.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        s.setup_solve(P, q, A, l, u, colmajor=True)  # first call of the shape: allocations happen here
        st.synchronize()
        x0 = s.solution()[0].copy()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            s.setup_solve(P, q, A, l, u, colmajor=True)
        for scale in (1.5, -0.5):
            q.mul_(scale)  # new linear term in the same buffers
            g.replay()
            st.synchronize()
            xg, yg = (a.copy() for a in s.solution()[:2])
            s.setup_solve(P, q, A, l, u, colmajor=True)
            st.synchronize()
            xd, yd = (a.copy() for a in s.solution()[:2])
            assert np.array_equal(xg, xd) and np.array_equal(yg, yd)
            assert not np.array_equal(xg, x0)
    assert s.kernel_name() == "wg2_16x8_7x7s_w2"  # (50,100): the stacked operator, wg_stack.hip


def test_stacked_and_padded_operator_of_the_c3_grid():
    """m <= 112, n <= 56 on the two-wave 16 x 8 grid: problems with m <= 104 run the iteration on the STACKED operator (10 tile rows,
    wg_stack.hip), the others on the padded one (11 rows) — both against the oracle at their limits, fixed and under termination"""
    from sqp_solver_amd.problems import random_qp_batch

    for (n, m, kern) in ((50, 100, "wg2_16x8_7x7s_w2"), (56, 104, "wg2_16x8_7x7s_w2"), (48, 103, "wg2_16x8_7x7s_w2"), (48, 105, "wg2_16x8_7x7_w2"), (33, 65, "wg2_16x8_7x7s_w2"),
                         (56, 112, "wg2_16x8_7x7_w2"), (50, 111, "wg2_16x8_7x7_w2")):
        cases.parity_fixed_iters(make_gpu, n, m, 16, iters=80)
        s = make_gpu(n, m, 4)
        s.setup_solve(*random_qp_batch(4, n, m, seed=1))
        assert s.kernel_name() == kern, (n, m, s.kernel_name())
        cases.parity_termination(make_gpu, n, m, 24, adaptive=True)
    cases.parity_termination(make_gpu, 49, 111, 24, sqp_settings=True)


@pytest.mark.parametrize("n,m,batch,adaptive_ok", [(2, 3, 5, False), (8, 12, 4, True), (20, 40, 3, True), (50, 100, 2, True)])
def test_api_sequence_fuzz(n, m, batch, adaptive_ok):
    """random call sequences with dispatch-changing settings in between, against per-QP oracle instances"""
    seen = set()
    for seed in range(1, 9):
        log, kernels = cases.api_sequence_fuzz(make_gpu, n, m, batch, seed=100 * n + seed, adaptive_ok=adaptive_ok)
        seen |= kernels
    print(n, m, sorted(seen))
    # the sequences did cross kernel families (verbose -> recording kernels, check_termination -> g16 / wg); the one-QP-per-lane
    # kernel records traces itself and serves every setting
    assert len(seen) >= (1 if n <= 4 else 2), seen


class _CsrFacade:
    """the dense-signature calls of cases.api_sequence_fuzz on the CSR entry points (A converted once per call)"""

    def __init__(self, s):
        self._s = s
        self.settings = s.settings

    def _csr(self, fn, P, q, A, l, u):
        rp, ci, v = cases.dense_to_csr(A)
        fn(P, q, rp, ci, v, l, u)

    def setup(self, *a): self._csr(self._s.setup_csr, *a)
    def update_qp(self, *a): self._csr(self._s.update_qp_csr, *a)
    def solve(self, *a): self._csr(self._s.solve_csr, *a)
    def setup_solve(self, *a): self._csr(self._s.setup_solve_csr, *a)
    def setup_solve_reuse(self, *a): self._csr(self._s.setup_solve_reuse_csr, *a)
    def set_state(self, *a): self._s.set_state(*a)
    def solution(self): return self._s.solution()
    def kernel_name(self): return self._s.kernel_name()


@pytest.mark.parametrize("n,m,batch", [(2, 3, 4), (20, 40, 3), (70, 150, 2)])
def test_api_sequence_fuzz_csr(n, m, batch):
    """the same random call sequences through the sparse entry points (native sparse kernel where the shape calls for it,
    expand + dense kernels otherwise)"""
    from sqp_solver_amd.problems import random_csr_qp_batch

    seen = set()
    for seed in range(1, 5):
        def make(n_, m_, b_, **kw):
            return _CsrFacade(make_gpu(n_, m_, b_, **kw))

        # a sparse A: the generator of the dense cases gives the other arrays, A is thinned out
        P, q, rp, ci, v, l, u, A = random_csr_qp_batch(batch, n, m, density=0.3 if n < 50 else 0.06, seed=seed)
        orig = cases.random_qp_batch
        cases.random_qp_batch = lambda b_, n_, m_, seed=0, **kw: (P, q, A, l, u)
        try:
            log, kernels = cases.api_sequence_fuzz(make, n, m, batch, seed=7 * n + seed)
        finally:
            cases.random_qp_batch = orig
        seen |= kernels
    print(n, m, sorted(seen))


@pytest.mark.parametrize("n,m", [(4, 4), (8, 12), (12, 24), (16, 24), (24, 40), (32, 64), (24, 96), (32, 128), (56, 112), (64, 128), (16, 224), (32, 224), (56, 224), (112, 32), (112, 64), (112, 128), (112, 208), (32, 448),
                                 (56, 448), (113, 209), (57, 449)])
def test_api_sequence_fuzz_at_kernel_shape_limits(n, m):
    """the largest (n, m) each compiled kernel shape takes (and one beyond the last: the fallback), through the random call sequences"""
    seen = set()
    for seed in (1, 2):
        log, kernels = cases.api_sequence_fuzz(make_gpu, n, m, 2, seed=1000 + 10 * n + seed, steps=8, adaptive_ok=n > 4)
        seen |= kernels
    print(n, m, sorted(seen))


def test_solver_calls_start_on_poisoned_lds():
    """tests/conftest.py fills the LDS of every CU with NaN before every call of the facade (tests/lds_poison.py): a kernel that reads a
    word of LDS it did not write — round 6's build_B of the 32- and 64-row grids — returns NaN here instead of passing or failing with the
    test order.  The (40, 132) sequence below is the one that found it: solve() on a LOADED factor, tile width 56 of 64 block columns."""
    import lds_poison

    active = lds_poison.STATE["lib"] is not None  # (off with SQPH_TEST_POISON_LDS=0, or where the helper could not be built: the session says so)
    before = lds_poison.STATE["calls"]
    cases.fused_then_solve(make_gpu, 40, 132, 2, adaptive=False)
    cases.fused_then_solve(make_gpu, 50, 300, 2, adaptive=False)
    if active:
        assert lds_poison.STATE["calls"] > before
