"""CPU tests of the boundary: libsqp_hip.so loads, exports every symbol include/sqp_hip.h declares,
its structs match the header, and it refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from sqp_solver_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "sqp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sqph_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _capi.load()
    names = header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(L, name), name
    assert sorted(_capi.SYMBOLS) == names
    assert L.sqph_version() == 1


def test_struct_layouts_and_defaults():
    assert ctypes.sizeof(_capi.Info) == 40
    assert ctypes.sizeof(_capi.Settings) == 72
    s = _capi.Settings()
    _capi.load().sqph_default_settings(ctypes.byref(s))
    # QPSolverSettings defaults, reference include/solvers/qp.hpp:38-53
    assert (s.rho, s.sigma, s.alpha, s.eps_rel, s.eps_abs) == (0.1, 1e-6, 1.0, 1e-3, 1e-3)
    assert (s.max_iter, s.check_termination, s.warm_start, s.adaptive_rho) == (1000, 25, 0, 0)
    assert (s.adaptive_rho_tolerance, s.adaptive_rho_interval, s.verbose) == (5.0, 25, 0)


def test_algorithmic_bytes_formula():
    L = _capi.load()
    assert L.sqph_algorithmic_bytes(20, 40, _capi.F64) == 10920
    assert L.sqph_algorithmic_bytes(50, 100, _capi.F64) == 63240  # C3


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    from sqp_solver_amd import QPSolverBatch, SqphError

    with pytest.raises(SqphError, match="no HIP device"):
        QPSolverBatch(2, 3, 1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sqp_solver_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "qp_oracle" not in txt, f


def test_shard_bounds():
    from sqp_solver_amd.dist import shard_bounds

    for total, world in ((65536, 8), (10, 4), (7, 8), (0, 3)):
        spans = [shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
