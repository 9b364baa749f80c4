"""CPU tests: the product's HIP kernels executed lane-for-lane by the host SIMT emulator
(tests/sim/) and compared with the oracle. These need no GPU."""
import numpy as np
import pytest

import cases
import simlib


def make_generic(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.GENERIC, nt=kw.get("nt", 64), legacy_cold_start=legacy_cold_start)


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


# ---------------------------------------------------------------- register-tiled wave-per-QP kernel
def make_tile(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.TILE, legacy_cold_start=legacy_cold_start)


@pytest.mark.parametrize("case", cases.REFERENCE_CASES, ids=lambda f: f.__name__)
def test_tile_reference_cases(case):
    case(make_tile)


@pytest.mark.parametrize("n,m", [(2, 3), (5, 7), (16, 24), (20, 40), (32, 64), (50, 100), (56, 104), (33, 9)])
def test_tile_parity_fixed(n, m):
    cases.parity_fixed_iters(make_tile, n, m, 2, iters=60 if n > 32 else 150)


def test_tile_parity_alpha_and_float():
    cases.parity_fixed_iters(make_tile, 9, 14, 3, iters=100, alpha=1.6)
    cases.parity_fixed_iters(make_tile, 20, 40, 2, iters=100, dtype=np.float32)


@pytest.mark.parametrize("kw", [dict(), dict(adaptive=True), dict(sqp_settings=True)], ids=["default", "adaptive", "sqp"])
@pytest.mark.parametrize("n,m", [(12, 20), (50, 100)])
def test_tile_parity_termination(n, m, kw):
    cases.parity_termination(make_tile, n, m, 3 if n > 20 else 6, **kw)


def test_tile_state_paths():
    cases.warm_start_and_resolve(make_tile)
    cases.set_state_warm_start(make_tile)
    cases.uninitialized_and_numerical_issues(make_tile)
    cases.shared_matrices(make_tile)
    cases.edge_shapes(make_tile)


# ---------------------------------------------------------------- workgroup-tiled kernel (matrices in VGPRs, vectors in LDS)
def make_wg(n, m, batch, dtype=np.float64, legacy_cold_start=False, **kw):
    return simlib.SimSolverBatch(n, m, batch, dtype=dtype, variant=simlib.WG, legacy_cold_start=legacy_cold_start)


@pytest.mark.parametrize("case", cases.REFERENCE_CASES, ids=lambda f: f.__name__)
def test_wg_reference_cases(case):
    case(make_wg)


@pytest.mark.parametrize("n,m", [(2, 3), (5, 7), (16, 24), (20, 40), (32, 64), (50, 100), (56, 112), (64, 128), (33, 9), (64, 120)])
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
