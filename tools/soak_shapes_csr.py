"""One-off soak (GPU): the API-sequence fuzz through the CSR entry points over random sparse shapes (n <= 224, m <= 512 native, beyond: expand)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cases
from test_gpu_parity import make_gpu, _CsrFacade
import lds_poison
print("LDS poison before every call:", lds_poison.install())
from sqp_solver_amd.problems import random_csr_qp_batch
rng = np.random.default_rng(777 + int(os.environ.get("SQPH_SOAK_SEED", "0")))
seen = {}; fails = []
for t in range(50):
    n = int(rng.integers(2, 240)); m = int(rng.integers(1, 530)); dens = float(rng.choice([0.03, 0.08, 0.3]))
    if n * m * dens > 9000: dens = 9000.0 / (n * m)
    P, q, rp, ci, v, l, u, A = random_csr_qp_batch(2, n, m, density=max(dens, 1.5 / n), seed=t)
    orig = cases.random_qp_batch
    cases.random_qp_batch = lambda b_, n_, m_, seed=0, **kw: (P, q, A, l, u)
    try:
        log, kernels = cases.api_sequence_fuzz(lambda n_, m_, b_, **kw: _CsrFacade(make_gpu(n_, m_, b_, **kw)), n, m, 2, seed=900 + t, steps=6, adaptive_ok=False)
        for k in kernels: seen[k] = seen.get(k, 0) + 1
    except AssertionError as e:
        fails.append((n, m, dens, str(e)[:300]))
    except Exception as e:
        fails.append((n, m, dens, "EXC " + repr(e)[:300]))
    finally:
        cases.random_qp_batch = orig
print("kernels exercised:", sorted(seen.items()))
print("failures:", len(fails))
for f in fails[:20]: print(f)
