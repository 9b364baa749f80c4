"""One-off soak (GPU): the API-sequence fuzz with the fp32 interface (QPSolver<float>: fp32 in and out, fp64 arithmetic) over random shapes."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cases
from test_gpu_parity import make_gpu
rng = np.random.default_rng(4242)
orig = cases.random_qp_batch
def rounded(*a, **k):
    return tuple(np.asarray(x, dtype=np.float32).astype(np.float64) for x in orig(*a, **k))
cases.random_qp_batch = rounded
cases.TOL_F64 = 2e-6   # outputs are rounded to fp32 at the interface (6e-8 relative per entry, max-norm relative error)
seen = {}; fails = []
for t in range(40):
    n = int(rng.integers(1, 70)); m = int(rng.integers(1, 460)) if rng.random() < 0.4 else int(rng.integers(1, 140))
    try:
        log, kernels = cases.api_sequence_fuzz(lambda n_, m_, b_, **kw: make_gpu(n_, m_, b_, dtype=np.float32, **kw), n, m, 2, seed=7000 + t, steps=6, adaptive_ok=False)
        for k in kernels: seen[k] = seen.get(k, 0) + 1
    except AssertionError as e:
        fails.append((n, m, str(e)[:300]))
    except Exception as e:
        fails.append((n, m, "EXC " + repr(e)[:300]))
print("kernels exercised:", sorted(seen.items()))
print("failures:", len(fails))
for f in fails[:20]: print(f)
