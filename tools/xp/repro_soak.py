"""re-run one case of tools/soak_shapes.py (SQPH_SOAK_SEED, n, m) with the full assertion text"""
import sys, os, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cases
from test_gpu_parity import make_gpu
rng = np.random.default_rng(12345 + int(os.environ.get("SQPH_SOAK_SEED", "0")))
N, M = int(sys.argv[1]), int(sys.argv[2])
for t in range(120):
    n = int(rng.integers(1, 70)); m = int(rng.integers(0, 460)) if rng.random() < 0.4 else int(rng.integers(0, 140))
    if m == 0 and n > 8: m = 1
    if (n, m) != (N, M): continue
    for fg in (False, True):
        mk = (lambda n_, m_, b_, **kw: make_gpu(n_, m_, b_, force_generic=fg, **kw))
        try:
            log, kernels = cases.api_sequence_fuzz(mk, n, m, 2, seed=5000 + t, steps=7, adaptive_ok=(n > 4 and n <= m <= 2.5 * n + 20))
            print("t", t, "force_generic", fg, "ok", kernels)
        except AssertionError:
            print("t", t, "force_generic", fg, "FAILED")
            traceback.print_exc()
