"""CPU tests: the product's HIP kernels executed lane-for-lane by the host SIMT emulator
(tests/sim/) and compared with the oracle. These need no GPU."""
import numpy as np
import pytest

import cases
import oracle
import simlib


def make_generic(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.GENERIC, nt=kw.get("nt", 64), legacy_cold_start=legacy_cold_start,
                                 keep_factor=kw.get("keep_factor", False))


@pytest.mark.parametrize("case", cases.REFERENCE_CASES, ids=lambda f: f.__name__)
def test_generic_reference_cases(case):
    case(make_generic)


@pytest.mark.parametrize("n,m,nt", [(5, 7, 64), (20, 40, 64), (20, 40, 256), (70, 130, 256)])
def test_generic_parity_fixed(n, m, nt):
    iters = 60 if n > 50 else 200
    cases.parity_fixed_iters(lambda *a, **k: make_generic(*a, nt=nt, **k), n, m, 2, iters=iters)


def test_generic_parity_alpha():
    cases.parity_fixed_iters(make_generic, 9, 14, 3, iters=100, alpha=1.6)


def test_generic_parity_float():
    cases.parity_fixed_iters(make_generic, 10, 15, 2, iters=100, dtype=np.float32)


@pytest.mark.parametrize("kw", [dict(), dict(adaptive=True), dict(sqp_settings=True)], ids=["default", "adaptive", "sqp"])
def test_generic_parity_termination(kw):
    cases.parity_termination(make_generic, 12, 20, 6, **kw)


def test_generic_warm_start():
    cases.warm_start_and_resolve(make_generic)
    cases.set_state_warm_start(make_generic)


def test_generic_status_paths():
    cases.uninitialized_and_numerical_issues(make_generic)


def test_generic_shared_and_edges():
    cases.shared_matrices(make_generic)
    cases.edge_shapes(make_generic)


# ---------------------------------------------------------------- workgroup-tiled kernel (matrices in VGPRs, vectors in LDS)
def make_wg(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.WG, legacy_cold_start=legacy_cold_start, keep_factor=kw.get("keep_factor", False))


@pytest.mark.parametrize("case", cases.REFERENCE_CASES, ids=lambda f: f.__name__)
def test_wg_reference_cases(case):
    case(make_wg)


@pytest.mark.parametrize("n,m", [(2, 3), (5, 7), (16, 24), (20, 40), (32, 64), (24, 96), (32, 128), (50, 100), (56, 112), (64, 128), (33, 9), (64, 120)])
def test_wg_parity_fixed(n, m):
    cases.parity_fixed_iters(make_wg, n, m, 2, iters=60 if n > 32 else 150)


def test_wg_parity_alpha_and_float():
    cases.parity_fixed_iters(make_wg, 9, 14, 3, iters=100, alpha=1.6)
    cases.parity_fixed_iters(make_wg, 20, 40, 2, iters=100, dtype=np.float32)
    cases.parity_fixed_iters(make_wg, 50, 100, 2, iters=60, dtype=np.float32)


@pytest.mark.parametrize("kw", [dict(), dict(adaptive=True), dict(sqp_settings=True)], ids=["default", "adaptive", "sqp"])
@pytest.mark.parametrize("n,m,b", [(12, 20, 6), (20, 40, 24), (50, 100, 3)])
def test_wg_parity_termination(n, m, b, kw):
    cases.parity_termination(make_wg, n, m, b, **kw)


def test_wg_state_paths():
    cases.warm_start_and_resolve(make_wg)
    cases.set_state_warm_start(make_wg)
    cases.uninitialized_and_numerical_issues(make_wg)
    cases.shared_matrices(make_wg)
    cases.edge_shapes(make_wg)
    cases.warm_start_and_resolve(make_wg, n=50, m=100, batch=2)


# ------------------------------------------------------------------ the sparse-A kernel (admm_csr_kernel.h), 1024 lanes per QP
def make_csr(n, m, batch, dtype=np.float64, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.CSR, **kw)


def test_csr_kernel_reference_cases():
    """tests/qp_solver_sparse_test.cpp (SimpleQP, multiple solve, update_qp) through the sparse kernel, tile edge 1"""
    cases.csr_reference_cases(make_csr)


@pytest.mark.parametrize("n,m,density,shared", [(12, 20, 0.3, False), (40, 60, 0.15, True)], ids=["t1", "t2"])
def test_csr_kernel_parity(n, m, density, shared):
    cases.csr_parity(make_csr, n, m, 1, iters=30, density=density, shared_pattern=shared)


def test_csr_kernel_adaptive_rho_refactors_in_kernel():
    from sqp_solver_amd.problems import random_csr_qp_batch

    n, m = 14, 24
    P, q, rp, ci, v, l, u, A = random_csr_qp_batch(2, n, m, density=0.3, seed=4)
    s = make_csr(n, m, 2)
    s.settings.adaptive_rho, s.settings.adaptive_rho_interval, s.settings.eps_abs, s.settings.eps_rel = 1, 10, 1e-5, 1e-5
    s.setup_solve_csr(P, q, rp, ci, v, l, u)
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings))
    assert (io["rho_updates"] > 1).any()  # the case does exercise a refactorisation
    assert (info.status == io["status"]).all() and (info.iter == io["iter"]).all() and (info.rho_updates == io["rho_updates"]).all()
    assert cases.relerr(x, xo) < cases.TOL_F64 and cases.relerr(y, yo) < cases.TOL_F64


# ------------------------------------------------------------------ the block-row sparse kernel (admm_csrb_kernel.h), 512 lanes per QP
def make_csrb(n, m, batch, dtype=np.float64, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.CSRB, **kw)


class _DenseAsCsr(simlib.SimSolverBatch):
    """the dense-A call surface on top of the CSR entry points (every entry of A stored, NaN included: the stateful cases of cases.py)"""

    @staticmethod
    def _csr(A):
        A = np.asarray(A, np.float64)
        lead = A.shape[:-2]
        m, n = A.shape[-2:]
        rp = np.broadcast_to(np.arange(m + 1, dtype=np.int32) * n, lead + (m + 1,)).copy()
        ci = np.broadcast_to(np.tile(np.arange(n, dtype=np.int32), m), lead + (m * n,)).copy()
        return rp, ci, A.reshape(lead + (m * n,))

    def setup(self, P, q, A, l, u):
        self.setup_csr(P, q, *self._csr(A), l, u)

    def update_qp(self, P, q, A, l, u):
        self.update_qp_csr(P, q, *self._csr(A), l, u)

    def solve(self, P, q, A, l, u):
        self.solve_csr(P, q, *self._csr(A), l, u)

    def setup_solve(self, P, q, A, l, u):
        self.setup_solve_csr(P, q, *self._csr(A), l, u)

    def update_solve(self, P, q, A, l, u):
        self.update_solve_csr(P, q, *self._csr(A), l, u)


def make_csrb_dense_A(n, m, batch, dtype=np.float64, **kw):
    return _DenseAsCsr(n, m, batch, dtype=dtype, variant=simlib.CSRB, **kw)


def test_csrb_kernel_reference_cases():
    """tests/qp_solver_sparse_test.cpp (SimpleQP, multiple solve, update_qp) through the block-row kernel, one block row"""
    cases.csr_reference_cases(make_csrb)


@pytest.mark.parametrize("n,m,density,shared", [(12, 20, 0.3, False), (30, 44, 0.2, False), (40, 60, 0.15, True), (70, 90, 0.1, False)],
                         ids=["nb1", "nb2", "nb3", "nb5"])
def test_csrb_kernel_parity(n, m, density, shared):
    """MFMA set-up with the blocks in registers, in-wavefront stage-1 reduction, register-resident slices of A: 1, 2, 3 and 5 block rows
    (5: every block-owning wavefront pair plus the single middle row)"""
    cases.csr_parity(make_csrb, n, m, 1, iters=30, density=density, shared_pattern=shared)


def test_csrb_kernel_config5_shape():
    """BASELINE config 5's shape (13 block rows, seven block-owning wavefronts and the eliminating one)"""
    cases.csr_parity(make_csrb, 200, 400, 1, iters=12, density=0.05)


def test_csrb_kernel_dense_rows_take_the_lds_products():
    """rows of 20 entries on 300 rows: more than KR entries per lane, the sparse products read their entries from LDS"""
    cases.csr_parity(make_csrb, 20, 300, 1, iters=20, density=1.0)


def test_csrb_kernel_adaptive_rho_refactors_in_kernel():
    from sqp_solver_amd.problems import random_csr_qp_batch

    n, m = 14, 24
    P, q, rp, ci, v, l, u, A = random_csr_qp_batch(2, n, m, density=0.3, seed=4)
    s = make_csrb(n, m, 2)
    s.settings.adaptive_rho, s.settings.adaptive_rho_interval, s.settings.eps_abs, s.settings.eps_rel = 1, 10, 1e-5, 1e-5
    s.setup_solve_csr(P, q, rp, ci, v, l, u)
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings))
    assert (io["rho_updates"] > 1).any()  # the case does exercise a refactorisation
    assert (info.status == io["status"]).all() and (info.iter == io["iter"]).all() and (info.rho_updates == io["rho_updates"]).all()
    assert cases.relerr(x, xo) < cases.TOL_F64 and cases.relerr(y, yo) < cases.TOL_F64


@pytest.mark.parametrize("n,m,density,sparse_P", [(30, 44, 0.2, False), (40, 60, 0.15, True)], ids=["nb2", "nb3_sp"])
def test_csrb_kernel_setup_solve_reuse(n, m, density, sparse_P):
    """MODE_SAME_MATRICES on the block-row kernel (sqph_setup_solve_reuse_csr / _csr_sp): the resident blocks are loaded instead of
    factorised where the freshly classified rho vector equals the stored one; a failed set-up refactors"""
    cases.soc_factor_reuse_csr(make_csrb, n=n, m=m, batch=2, density=density, sparse_P=sparse_P)


def test_csr_kernel_setup_solve_reuse():
    """... and on the 32 x 32 lane-grid sparse kernel"""
    cases.soc_factor_reuse_csr(make_csr, n=14, m=24, batch=2, density=0.3)


def test_csrb_kernel_stateful_calls():
    cases.fused_then_solve(make_csrb_dense_A, n=33, m=40, batch=2)
    cases.warm_start_and_resolve(make_csrb_dense_A, n=20, m=30)
    cases.uninitialized_and_numerical_issues(make_csrb_dense_A)


# ------------------------------------------------------------------ four QPs per wavefront (admm_wg_kernel.h, run_group)
def make_g16(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.G16, legacy_cold_start=legacy_cold_start, keep_factor=kw.get("keep_factor", False))


@pytest.mark.parametrize("case", cases.REFERENCE_CASES, ids=lambda f: f.__name__)
def test_g16_reference_cases(case):
    case(make_g16)


@pytest.mark.parametrize("n,m,batch", [(2, 3, 7), (5, 7, 5), (8, 12, 6), (12, 24, 5), (11, 21, 3)])
def test_g16_parity_fixed(n, m, batch):
    """batches that are not multiples of four leave groups of the last wavefront empty"""
    cases.parity_fixed_iters(make_g16, n, m, batch, iters=100)


def test_g16_parity_alpha_and_float():
    cases.parity_fixed_iters(make_g16, 9, 14, 3, iters=100, alpha=1.6)
    cases.parity_fixed_iters(make_g16, 10, 15, 2, iters=100, dtype=np.float32)


@pytest.mark.parametrize("kw", [dict(), dict(adaptive=True), dict(sqp_settings=True)], ids=["default", "adaptive", "sqp"])
@pytest.mark.parametrize("n,m", [(12, 20), (7, 11)])
def test_g16_parity_termination(n, m, kw):
    """the four QPs of a wavefront stop / refactor at different iterations"""
    cases.parity_termination(make_g16, n, m, 6, **kw)


def test_g16_state_paths():
    cases.warm_start_and_resolve(make_g16)
    cases.set_state_warm_start(make_g16)
    cases.uninitialized_and_numerical_issues(make_g16)
    cases.shared_matrices(make_g16)
    cases.edge_shapes(make_g16)


def test_wg_four_wave_large_tile_shape():
    """the m <= 224, 56 < n <= 112 shape (32 x 16 lanes, eight waves) under the emulator"""
    cases.parity_fixed_iters(lambda n, m, b, **kw: simlib.SimSolverBatch(n, m, b, variant=simlib.WG), 70, 150, 1, iters=25)


@pytest.mark.parametrize("n,m", [(100, 20), (112, 32), (90, 60), (80, 120)])
def test_wg_wide_shapes(n, m):
    """56 < n <= 112 with few constraints: the 16 x 16 grids with 2 / 4 / 8 tile rows"""
    mk = lambda n_, m_, b, **kw: simlib.SimSolverBatch(n_, m_, b, variant=simlib.WG, keep_factor=kw.get("keep_factor", False))  # noqa: E731
    cases.parity_fixed_iters(mk, n, m, 1, iters=20)
    if n <= 100:
        cases.fused_then_solve(mk, n, m, 1, adaptive=False)


@pytest.mark.parametrize("n,m", [(10, 150), (16, 224), (30, 200), (50, 140), (56, 224), (20, 400), (32, 448), (50, 300)])
def test_wg_tall_shapes(n, m):
    """the 32 x 8 and 64 x 8 lane grids (many more constraints than variables: m <= 224 with n <= 16 / 32 / 56, m <= 448 with
    n <= 32 / 56) under the emulator"""
    mk = lambda n_, m_, b, **kw: simlib.SimSolverBatch(n_, m_, b, variant=simlib.WG, keep_factor=kw.get("keep_factor", False))  # noqa: E731
    cases.parity_fixed_iters(mk, n, m, 2, iters=25)
    if n <= 30:
        cases.parity_termination(mk, n, m, 2)
        # adaptive rho on these constraint-heavy QPs drives rho up: iterates stay inside the bar, the absolute floor of the reported
        # residuals (calibrated on m = 2n problems) does not apply
        # (n < 20: a batch of 20 with an explicit 15 % cap on the widened bar instead of the old one-QP floor of a batch of 2 —
        # with m = 14 n .. 15 n the adapted rho puts 1 of 20 QPs at (16, 224) and 2 of 20 at (10, 150) at 10x their own fp64 noise
        # floor; counted in the session summary like every hatch)
        cases.parity_termination(mk, n, m, 20 if n < 20 else 2, adaptive=True, diagnostics=False, **({"max_hatch_frac": 0.15} if n < 20 else {}))
    # every shape: solve() on a LOADED factor — n = 50 and 56 are the tiles whose width is not a multiple of the 16-column blocks of
    # the MFMA set-up (round 6: build_B read the never-written pad columns of the staged A block, NaN under the emulator's LDS poison
    # and on the GPU whenever the CU's previous workgroup had left NaN / inf there)
    cases.fused_then_solve(mk, n, m, 2, adaptive=False)  # (adaptive rho on these constraint-heavy QPs sits at the fp64 noise floor)


@pytest.mark.parametrize("make,n,m", [(make_generic, 6, 9), (make_wg, 8, 12), (make_wg, 50, 100), (make_g16, 8, 12), (make_wg, 20, 40)],
                         ids=["generic", "wg1", "wg2", "g16", "wg1_c2"])
def test_fused_call_then_solve(make, n, m):
    """factor residency policy (capi.hip, mirrored by simlib): fused setup_solve without / with keep_factor, then solve()"""
    cases.fused_then_solve(make, n=n, m=m, batch=2)
    cases.solve_with_other_P(make, n=n, m=m, batch=2)  # solve(qp) with another P than setup(qp): residuals from the call's P


# ---------------------------------------------------------------- one QP per lane (tiny problems)
def make_lane(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.LANE, legacy_cold_start=legacy_cold_start, keep_factor=kw.get("keep_factor", False))


@pytest.mark.parametrize("case", [c for c in cases.REFERENCE_CASES if c.__name__ != "ref_legacy_TestConstraint"], ids=lambda f: f.__name__)
def test_lane_reference_cases(case):
    case(make_lane)  # (the 5 x 5 legacy constraint case is beyond the lane kernel's shapes)


@pytest.mark.parametrize("n,m,batch", [(2, 3, 70), (1, 1, 5), (2, 1, 9), (3, 3, 66), (4, 6, 130), (3, 0, 4)])
def test_lane_parity_fixed(n, m, batch):
    if m == 0:
        cases.edge_shapes(make_lane, shapes=((1, 1), (3, 0), (4, 1), (2, 3)))
        return
    cases.parity_fixed_iters(make_lane, n, m, batch, iters=150, dual_floor=True)


def test_lane_parity_alpha_float_termination_and_state_paths():
    cases.parity_fixed_iters(make_lane, 2, 3, 9, iters=100, alpha=1.6)
    cases.parity_fixed_iters(make_lane, 4, 6, 5, iters=100, dtype=np.float32)
    for kw in (dict(), dict(adaptive=True), dict(sqp_settings=True)):
        # (adaptive rho on QPs this small: the reported residuals are compared on the QPs whose reference diagnostics are reproducible)
        cases.parity_termination(make_lane, 4, 6, 40, diagnostics=True if not kw else "stable", **kw)
        cases.parity_termination(make_lane, 2, 3, 40, diagnostics=True if not kw else "stable", **kw)
    cases.warm_start_and_resolve(make_lane, n=4, m=6)
    cases.set_state_warm_start(make_lane, n=3, m=5)
    cases.uninitialized_and_numerical_issues(make_lane)
    cases.shared_matrices(make_lane, n=4, m=6)
    cases.fused_then_solve(make_lane, n=4, m=5, batch=3)
    for kind in ("all_eq", "half_eq", "illcond"):  # (the adaptive kinds are chaotic on QPs this small: see parity_termination)
        cases.stress_parity(make_lane, 4, 6, 8, kind, iters=120)


def make_quad(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.LANE_QUAD, legacy_cold_start=legacy_cold_start, keep_factor=kw.get("keep_factor", False))


@pytest.mark.parametrize("case", [c for c in cases.REFERENCE_CASES if c.__name__ != "ref_legacy_TestConstraint"], ids=lambda f: f.__name__)
def test_quad_reference_cases(case):
    case(make_quad)


def test_quad_variant_of_the_lane_kernel():
    """LaneKernel<..., LPQ = 4>: four lanes per QP (m <= 4), quads of a wavefront diverging freely — fixed iterations at the exact
    shapes and a padded one, termination in the three settings (iteration counts differ between the quads of a wave), state paths"""
    for (n, m, b) in ((2, 3, 37), (2, 2, 20), (3, 3, 18), (2, 1, 5), (4, 4, 9), (3, 2, 7)):
        cases.parity_fixed_iters(make_quad, n, m, b, iters=120)
    cases.parity_fixed_iters(make_quad, 2, 3, 8, iters=100, alpha=1.6)
    cases.parity_fixed_iters(make_quad, 2, 3, 8, iters=100, dtype=np.float32)
    for kw in (dict(), dict(adaptive=True), dict(sqp_settings=True)):
        cases.parity_termination(make_quad, 2, 3, 40, diagnostics=True if not kw else "stable", **kw)
        cases.parity_termination(make_quad, 3, 3, 24, diagnostics=True if not kw else "stable", **kw)
    cases.warm_start_and_resolve(make_quad, n=3, m=3)
    cases.set_state_warm_start(make_quad, n=3, m=4)
    cases.uninitialized_and_numerical_issues(make_quad, n=3, m=3)
    cases.shared_matrices(make_quad, n=4, m=4)
    cases.fused_then_solve(make_quad, n=3, m=3, batch=3)
    cases.soc_factor_reuse(make_quad, n=2, m=3, batch=6)
    cases.solve_with_other_P(make_quad, n=2, m=3, batch=5)


def make_lane_f32(n, m, batch, dtype=np.float32, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=np.float32, variant=simlib.LANE_F32, legacy_cold_start=legacy_cold_start)


def test_lane_true_fp32_variant():
    """SQPH_FLAG_F32_ARITH: iterates, factor and residuals in fp32.  Stated tolerance: within TOL_F32 = 5e-3 of the
    reference's QPSolver<float> (the float oracle), and no further from the fp64 solution of the same float-valued problem than
    4x the float oracle is (floor 2e-3); status and iteration counts equal to the float oracle's under default termination."""
    for (n, m) in ((2, 3), (4, 6), (3, 3)):
        ex, ey, ez = cases.parity_fixed_iters(make_lane_f32, n, m, 128, iters=150, dtype=np.float32, dual_floor=True, f32_floor=2e-3)
        assert ex < 2e-3 and ey < 2e-3, (ex, ey)
    from sqp_solver_amd.problems import random_qp_batch

    P, q, A, l, u = random_qp_batch(128, 4, 6, seed=7, dtype=np.float32)
    s = make_lane_f32(4, 6, 128)
    s.setup_solve(P, q, A, l, u)
    x, y, z, info = s.solution()
    xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings), dtype=np.float32)
    assert (info.status == io["status"]).all() and (info.iter == io["iter"]).all()
    assert cases.relerr(x, xo) < cases.TOL_F32
    cases.ref_testSinglePrecisionFloat(make_lane_f32)


def make_wg_stack(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.WG_STACK, legacy_cold_start=legacy_cold_start, keep_factor=kw.get("keep_factor", False))


def test_wg_stacked_operator():
    """WgKernel::run<CHECKS, false, STACK = true> (wg_stack.hip): the rows of W' follow the m rows of B in one stacked operator of
    TR + TW - 1 tile rows — W' row j at stacked row 104 + j (inside the last B tile row), m at the limit of 104 and just below,
    tiny ones; termination in the three settings; state paths"""
    for (n, m, b) in ((50, 100, 2), (56, 104, 2), (48, 103, 1), (33, 65, 2), (7, 3, 2)):
        cases.parity_fixed_iters(make_wg_stack, n, m, b, iters=60)
    cases.parity_fixed_iters(make_wg_stack, 49, 99, 2, iters=40, dtype=np.float32)
    for kw in (dict(), dict(adaptive=True), dict(sqp_settings=True)):
        cases.parity_termination(make_wg_stack, 50, 100, 3, **kw)
    cases.fused_then_solve(make_wg_stack, n=50, m=100, batch=2)
    cases.warm_start_and_resolve(make_wg_stack, n=40, m=77)
    cases.soc_factor_reuse(make_wg_stack, n=50, m=100, batch=2)
    cases.failing_pivots(make_wg_stack, n=50, m=100, batch=4)  # the MFMA set-up: the pivot flag of a diagonal block, NaN through the block products
    cases.failing_pivots(make_wg, n=8, m=12, batch=4)
    cases.failing_pivots(make_wg, n=20, m=40, batch=4)
    cases.failing_pivots(make_wg, n=60, m=120, batch=3)  # the MFMA set-up of the four-wave grid (P read from global memory)
    cases.failing_pivots(make_wg, n=90, m=60, batch=3)  # ... with seven block columns in swizzled blocks (56 < n <= 112)


def test_generic_blocked_setup():
    """the generic kernel's blocked MFMA set-up on global-memory blocks (admm_generic_msetup.h: n >= 48, >= 4 wavefronts): fixed
    iterations, termination with adaptive rho (in-kernel refactorisation), a non-SPD Schur complement, solve() on a kept factor"""
    def mk(nt):
        return lambda n, m, b, **kw: simlib.SimSolverBatch(n, m, b, variant=simlib.GENERIC, nt=nt, keep_factor=kw.get("keep_factor", False))
    cases.parity_fixed_iters(mk(256), 60, 80, 2, iters=30)
    cases.parity_fixed_iters(mk(512), 100, 50, 1, iters=20)
    cases.parity_fixed_iters(mk(256), 49, 7, 2, iters=20)
    cases.parity_termination(mk(256), 70, 120, 2, adaptive=True)
    cases.fused_then_solve(mk(256), n=64, m=40, batch=2)
    P, q, A, l, u = cases.random_qp_batch(2, 60, 80, seed=23)
    P = P.copy()
    P[0] = -100.0 * np.eye(60)  # S = P + sigma I + A'RA indefinite
    s = mk(256)(60, 80, 2)
    s.settings.max_iter, s.settings.check_termination = 20, 0
    s.setup_solve(P, q, A, l, u)
    x, y, z, info = s.solution()
    assert info.status[0] == cases.NUMERICAL_ISSUES and info.status[1] == cases.MAX_ITER_EXCEEDED


def test_csr_dense_tile_edge_8():
    """the CU-wide kernel's dense mode at tile edge 8 (224 < n <= 256, round 6) under the emulator: fixed iterations and the default
    termination with adaptive rho at (250, 300) — the shape whose checking instantiation the GPU miscompiles with
    -structurizecfg-skip-uniform-regions (sqp_solver_amd/build.py): the emulator says the source is right"""
    mk = lambda n, m, b, **kw: simlib.SimSolverBatch(n, m, b, variant=11, keep_factor=kw.get("keep_factor", False))  # noqa: E731
    cases.parity_fixed_iters(mk, 250, 300, 1, iters=10)
    cases.parity_termination(mk, 250, 300, 1, adaptive=True)


def make_csr_dense(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.CSR_DENSE, legacy_cold_start=legacy_cold_start, keep_factor=kw.get("keep_factor", False))


def test_cu_wide_kernel_dense_mode():
    """CsrKernel::run<CHECKS, DENSE = true> (csr_dense.hip): the factor in the workgroup's registers, A streamed from global memory —
    fixed iterations, termination (plain / adaptive rho / SQP settings), float interface, state paths; tile edges 1, 2 and 4"""
    for (n, m, b) in ((20, 40, 2), (40, 70, 2), (70, 30, 1), (100, 3, 1)):
        cases.parity_fixed_iters(make_csr_dense, n, m, b, iters=30)
    cases.parity_fixed_iters(make_csr_dense, 24, 50, 2, iters=30, dtype=np.float32)
    for kw in (dict(), dict(adaptive=True), dict(sqp_settings=True)):
        cases.parity_termination(make_csr_dense, 30, 61, 3, **kw)
    cases.fused_then_solve(make_csr_dense, n=33, m=40, batch=2)
    cases.warm_start_and_resolve(make_csr_dense, n=20, m=30)
    cases.uninitialized_and_numerical_issues(make_csr_dense)


def make_wg_f32(n, m, batch, dtype=np.float32, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=np.float32, variant=simlib.WG_F32, legacy_cold_start=legacy_cold_start, keep_factor=kw.get("keep_factor", False))


def test_wg_fp32_product_variant():
    """SQPH_FLAG_F32_ARITH at the BASELINE dense shapes (reference src/qp.cpp:385-386): B / W' tiles, operand
    vectors and partial sums of the two iteration stages in fp32 (v_pk_fma_f32), factorisation / iterates / residual checks in
    fp64.  Stated accuracy, without a floor (cases.parity_fixed_iters): x within TOL_F32 = 5e-3 of the reference's QPSolver<float>
    (float oracle); against the fp64 solution of the same float-valued problem x and z no further than 3x the float oracle's error
    (or 5e-6, ~40 fp32 ulps: a maximum over 3-12 QPs is noisy), y no further than 8x with no floor on these small batches (the GPU test states 2.5x / 6x on batches of 32-512: tests/test_gpu_parity.py).
    Default termination: status equal,
    iteration counts equal to the FP64 oracle's on at least 3 of 4 QPs (the stop test sits on an fp32-noisy residual)."""
    from sqp_solver_amd.problems import random_qp_batch

    for (n, m, b) in ((20, 40, 12), (50, 100, 6), (30, 60, 4), (56, 112, 3)):
        ex, ey, ez = cases.parity_fixed_iters(make_wg_f32, n, m, b, iters=150, dtype=np.float32, f32_floor=(5e-6, 0.0, 5e-6), f32_ratio=(3.0, 8.0, 3.0))
        assert ex < 2e-5 and ey < 1e-3, (n, m, ex, ey)
    for (n, m, b) in ((20, 40, 12), (50, 100, 6)):
        P, q, A, l, u = random_qp_batch(b, n, m, seed=3, dtype=np.float32)
        s = make_wg_f32(n, m, b)
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        f64 = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
        xo, yo, zo, io = oracle.solve_batch(f64(P), f64(q), f64(A), f64(l), f64(u), cases.oracle_settings(s.settings))
        assert (info.status == io["status"]).all()
        assert (info.iter == io["iter"]).mean() >= 0.75, (info.iter, io["iter"])
        same = info.iter == io["iter"]
        assert cases.relerr(x[same], xo[same]) < 1e-4
        # solve() on the resident factor == the fused call, bit for bit, at a fixed iteration count (under termination the fused
        # call keeps A x by recurrence from the fp32 products, a solve() on retained iterates streams A: the stop test may differ)
        s.settings.max_iter, s.settings.check_termination = 40, 0
        s.setup_solve(P, q, A, l, u)
        x, y, z, info = s.solution()
        s2 = make_wg_f32(n, m, b, keep_factor=True)
        s2.settings.max_iter, s2.settings.check_termination = 40, 0
        s2.setup(P, q, A, l, u)
        s2.solve(P, q, A, l, u)
        x2, y2, z2, info2 = s2.solution()
        assert np.array_equal(x2, x) and np.array_equal(y2, y) and np.array_equal(info2.iter, info.iter)
    cases.ref_testSinglePrecisionFloat(make_wg_f32)


@pytest.mark.parametrize("make,n,m", [(make_lane, 2, 3), (make_lane, 4, 6), (make_wg, 8, 12), (make_wg, 50, 100), (make_g16, 8, 12), (make_generic, 6, 9)],
                         ids=["lane2x3", "lane4x6", "wg1", "wg2", "g16", "generic"])
def test_setup_solve_reuse(make, n, m):
    """sqph_setup_solve_reuse (SOC): bit-identical to a plain setup+solve whether the factor is reused (lane, wg), rebuilt because
    rho moved, or the kernel ignores the hint (g16, generic)"""
    cases.soc_factor_reuse(make, n=n, m=m, batch=3)


@pytest.mark.parametrize("make,n,m", [(make_lane, 2, 3), (make_lane, 4, 6), (make_wg, 8, 12), (make_wg, 50, 100)], ids=["lane2x3", "lane4x6", "wg1", "wg2"])
def test_setup_solve_reuse_after_failed_setup(make, n, m):
    """the SOC fast path must not iterate on the factor of a set-up that ended in NUMERICAL_ISSUES"""
    cases.soc_reuse_after_failed_setup(make, n=n, m=m, batch=3)


def test_lane_golden_fixtures():
    """the config-4 golden fixtures (tests/golden/c4_*.npz) through the one-QP-per-lane kernel under the emulator"""
    import golden_io

    seen = 0
    for name, g in golden_io.load_all():
        if not name.startswith("c4_"):
            continue
        seen += 1
        s = make_lane(g["n"], g["m"], g["P"].shape[0])
        golden_io.apply_settings(s.settings, g)
        s.setup_solve(g["P"], g["q"], g["A"], g["l"], g["u"])
        x, y, z, info = s.solution()
        assert cases.relerr(x, g["x"]) < cases.TOL_F64 and cases.relerr1(y, g["y"]) < cases.TOL_F64, name
        assert (info.status == g["status"]).all() and (info.iter == g["iter"]).all() and (info.rho_updates == g["rho_updates"]).all(), name
    assert seen == 3


@pytest.mark.parametrize("n,m,density,pdens,shared", [(12, 20, 0.3, 0.3, False), (40, 60, 0.15, 0.1, True), (70, 90, 0.1, 0.05, False),
                                                       (200, 400, 0.05, 0.03, False)], ids=["nb1", "nb3", "nb5", "nb13"])
def test_csrb_kernel_sparse_P_is_bit_identical_to_dense_P(n, m, density, pdens, shared):
    """the block-row kernel's sparse-P instantiations (P in compressed columns: S phase and dual residual read it in place) against its
    dense-P ones on the matrix the columns encode: same bits, fixed iterations and under termination checks with adaptive rho; a
    column without a stored diagonal entry; the oracle"""
    from sqp_solver_amd.problems import random_csr_qp_batch

    B = 1 if n >= 200 else 2
    _, q, rp, ci, v, l, u, A = random_csr_qp_batch(B, n, m, density=density, seed=41, shared_pattern=shared)
    P = cases.sparse_spd(B, n, pdens, seed=42, shared_pattern=shared)
    cp, ri, pv = cases.dense_to_csr(P)
    if shared:
        rp, ci, cp, ri = rp[0], ci[0], cp[0], ri[0]
    for fixed in ((True, False) if n in (12, 70) else (True,)):
        outs = []
        for Parg in (P, (cp, ri, pv)):
            s = make_csrb(n, m, B)
            if fixed:
                s.settings.max_iter, s.settings.check_termination = (6 if n >= 200 else 25), 0
            else:
                s.settings.adaptive_rho, s.settings.adaptive_rho_interval, s.settings.eps_abs, s.settings.eps_rel = 1, 10, 1e-4, 1e-4
            s.setup_solve_csr(Parg, q, rp, ci, v, l, u)
            outs.append(s.solution())
        (x0, y0, z0, i0), (x1, y1, z1, i1) = outs
        assert np.array_equal(x0, x1) and np.array_equal(y0, y1) and np.array_equal(z0, z1)
        assert (i0.iter == i1.iter).all() and (i0.status == i1.status).all() and (i0.rho_updates == i1.rho_updates).all()
        assert np.array_equal(i0.res_prim, i1.res_prim) and np.array_equal(i0.res_dual, i1.res_dual)
        xo, yo, zo, io = oracle.solve_batch(P, q, A, l, u, cases.oracle_settings(s.settings))
        assert cases.relerr(x1, xo) < cases.TOL_F64 and cases.relerr(y1, yo) < cases.TOL_F64 and (i1.iter == io["iter"]).all()
    if n == 12:  # P = 0 with no stored entry at all (S = sigma I + A'RA), and the stateful calls
        z = (np.zeros(n + 1, np.int32), np.zeros(1, np.int32), np.zeros(1))
        a, b = make_csrb(n, m, B), make_csrb(n, m, B)
        for s, Parg in ((a, np.zeros((n, n))), (b, z)):
            s.settings.max_iter = 25
            s.setup_csr(Parg, q, rp, ci, v, l, u)
            s.solve_csr(Parg, q, rp, ci, v, l, u)
        assert np.array_equal(a.solution()[0], b.solution()[0]) and np.array_equal(a.solution()[1], b.solution()[1])


@pytest.mark.parametrize("n,m", [(4, 4), (8, 12), (16, 24), (32, 64), (32, 128), (50, 100), (56, 112), (64, 128), (16, 224), (32, 224), (56, 224),
                                 (112, 32), (112, 208), (32, 448), (56, 448)])
def test_wg_loaded_factor_at_every_shape_limit(n, m):
    """solve() and setup_solve_reuse() on a factor LOADED from the workspace, at the largest (n, m) of every compiled register-tiled shape,
    under the emulator's NaN-poisoned LDS: the path on which round 6's stale-LDS read lived (tile width 56 of the 64 columns of the MFMA
    set-up's blocks) — a factorisation leaves finite numbers where a loaded factor leaves whatever was there"""
    mk = lambda n_, m_, b, **kw: simlib.SimSolverBatch(n_, m_, b, variant=simlib.WG, keep_factor=kw.get("keep_factor", False))  # noqa: E731
    cases.fused_then_solve(mk, n, m, 2, adaptive=False)
    if (n, m) in ((50, 100), (56, 224), (56, 448), (112, 208)):  # (the SOC re-solve: the widest shape of each set-up variant; the GPU suite runs it everywhere)
        cases.soc_factor_reuse(mk, n, m, 2)
