"""One-off soak (GPU): the API-sequence fuzz of tests/cases.py over 120 random shapes (n < 70, m < 460; SQPH_SOAK_NMAX / _MMAX / _SHAPES widen it) — every kernel family and shape the
dispatch can pick.  Adaptive rho only where the problem is neither tiny nor wide (an iterate that converges exactly leaves residuals at
rounding level, and the reference's rho estimate is then noise: tests/cases.py::parity_termination)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cases
from test_gpu_parity import make_gpu
import lds_poison
print("LDS poison before every call:", lds_poison.install())
rng = np.random.default_rng(12345 + int(os.environ.get("SQPH_SOAK_SEED", "0")))
seen = {}
fails = []
NMAX, MMAX, SHAPES = int(os.environ.get("SQPH_SOAK_NMAX", "70")), int(os.environ.get("SQPH_SOAK_MMAX", "460")), int(os.environ.get("SQPH_SOAK_SHAPES", "120"))
for t in range(SHAPES):
    n = int(rng.integers(1, NMAX)); m = int(rng.integers(0, MMAX)) if rng.random() < 0.4 else int(rng.integers(0, min(MMAX, 2 * NMAX)))
    if m == 0 and n > 8: m = 1
    try:
        log, kernels = cases.api_sequence_fuzz(make_gpu, n, m, 2, seed=5000 + t, steps=7, adaptive_ok=(n > 4 and n <= m <= 2.5 * n + 20))
        for k in kernels: seen[k] = seen.get(k, 0) + 1
    except AssertionError as e:
        fails.append((n, m, t, str(e)[:1500]))
    except Exception as e:
        fails.append((n, m, "EXC " + repr(e)[:300]))
print("kernels exercised:", sorted(seen.items()))
print("failures:", len(fails))
for f in fails[:20]: print(f)
