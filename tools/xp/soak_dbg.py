"""tools/soak_shapes.py over cases T0..T1 only, against the library named by SQPH_XP_LIB (default: the shipped one)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
if os.environ.get("SQPH_XP_LIB"):
    import sqp_solver_amd.build as _b
    _b.LIB = os.path.abspath(os.environ["SQPH_XP_LIB"])
    _b.needs_build = lambda: False
import cases
from test_gpu_parity import make_gpu
rng = np.random.default_rng(12345 + int(os.environ.get("SQPH_SOAK_SEED", "0")))
T0, T1 = int(sys.argv[1]), int(sys.argv[2])
nf = 0
for t in range(120):
    n = int(rng.integers(1, 70)); m = int(rng.integers(0, 460)) if rng.random() < 0.4 else int(rng.integers(0, 140))
    if m == 0 and n > 8: m = 1
    if t < T0 or t > T1: continue
    try:
        log, kernels = cases.api_sequence_fuzz(make_gpu, n, m, 2, seed=5000 + t, steps=7, adaptive_ok=(n > 4 and n <= m <= 2.5 * n + 20))
    except AssertionError as e:
        nf += 1
        print("FAIL", n, m, t, str(e)[:600])
print("lib", os.environ.get("SQPH_XP_LIB", "shipped"), "range", T0, T1, "failures", nf)
