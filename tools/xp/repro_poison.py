"""one soak case (SQPH_SOAK_SEED, t) alone in the process, LDS of every CU poisoned before every solver call"""
import sys, os, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cases, torch
from test_gpu_parity import make_gpu
PL = ctypes.CDLL(os.path.abspath("xp_libs/lds_poison.so"))
pat = int(os.environ.get("PAT", "0xFFFFFFFF"), 16)
nbytes = int(os.environ.get("PBYTES", str(160 * 1024)))
def poison():
    torch.cuda.synchronize()
    rc = PL.lds_poison(ctypes.c_uint(pat), nbytes)
    assert rc == 0, rc
def mk(n, m, b, **kw):
    s = make_gpu(n, m, b, **kw)
    if os.environ.get("NOPOISON"): return s
    for name in ("setup", "update_qp", "solve", "setup_solve", "setup_solve_reuse"):
        f = getattr(s, name)
        def w(*a, _f=f, **k):
            poison()
            return _f(*a, **k)
        setattr(s, name, w)
    return s
rng = np.random.default_rng(12345 + int(os.environ.get("SQPH_SOAK_SEED", "0")))
T = [int(x) for x in sys.argv[1:]]
for t in range(120):
    n = int(rng.integers(1, 70)); m = int(rng.integers(0, 460)) if rng.random() < 0.4 else int(rng.integers(0, 140))
    if m == 0 and n > 8: m = 1
    if t not in T: continue
    try:
        log, kernels = cases.api_sequence_fuzz(mk, n, m, 2, seed=5000 + t, steps=7, adaptive_ok=(n > 4 and n <= m <= 2.5 * n + 20))
        print("ok", n, m, t, kernels)
    except AssertionError as e:
        print("FAIL", n, m, t, str(e)[:400])
