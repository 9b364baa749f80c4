"""quick GPU parity run of the C3-shape kernels of an experiment build (SQPH_LIB=...): fixed / termination / state paths vs the oracle"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, cases
from sqp_solver_amd import QPSolverBatch
def mk(n, m, b, dtype=np.float64, **kw):
    return QPSolverBatch(n, m, b, dtype=dtype, device=0, keep_factor=kw.get("keep_factor", False), legacy_cold_start=kw.get("legacy_cold_start", False))
for (n, m) in ((50, 100), (56, 104), (33, 65), (49, 99)):
    s = mk(n, m, 4); s.setup_solve(*cases.random_qp_batch(4, n, m, seed=1)); print("kernel", n, m, s.kernel_name())
    print(" fixed", cases.parity_fixed_iters(mk, n, m, 64, iters=200))
    if n == 49: print(" fixed f32 interface", cases.parity_fixed_iters(mk, n, m, 16, iters=100, dtype=np.float32))
    for kw in (dict(), dict(adaptive=True), dict(sqp_settings=True)):
        info = cases.parity_termination(mk, n, m, 64, **kw); print(" term", kw, cases.HATCH_COUNTS[-1]["excused"], cases.HATCH_COUNTS[-1]["widened"], int(info.iter.mean()))
cases.fused_then_solve(mk, n=50, m=100, batch=5); cases.soc_factor_reuse(mk, n=50, m=100, batch=5); cases.soc_reuse_after_failed_setup(mk, n=50, m=100, batch=5)
cases.warm_start_and_resolve(mk, n=50, m=100); cases.shared_matrices(mk); print("state paths ok")
