"""Debug helper (GPU): per-phase tick breakdown of the block-row sparse kernel (admm_csrb_kernel.h) from a -DSQPH_PHASE_TIMING build
(SQPH_LIB=<that library>; tools/slim_build.sh <out.so> -DSQPH_SLIM_CSR -DSQPH_PHASE_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_csr_qp_batch
n, m, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = 200
P, q, rp, ci, v, l, u, A = random_csr_qp_batch(8, n, m, density=0.05, seed=1)
rep = (B + 7) // 8
tile = lambda a: np.concatenate([a] * rep)[:B]
s = QPSolverBatch(n, m, B)
s.settings.max_iter = iters
s.settings.check_termination = 0
if os.environ.get("SQPH_PT_CHECKS"):  # the checking instantiation: a check every 25 iterations that never passes
    s.settings.check_termination = 25
    s.settings.eps_abs = s.settings.eps_rel = 1e-300
args = [tile(a) for a in (P, q, rp, ci, v, l, u)]
s.setup_solve_csr(*args)
s.enable_timing(True)
s.setup_solve_csr(*args)
ms = s.collect_kernel_ms()
x, y, z, info = s.solution()
if os.environ.get("SQPH_PT_SETUP") == "1":  # a -DSQPH_PT_SETUP=1 build: slots 3-6, 8 = sub-phases of the S phase
    sub = {3: "S: clear panel", 4: "S: A'RA accumulate", 5: "S: P add + prefetch", 6: "S: barrier waits", 8: "S: pick-up"}
elif os.environ.get("SQPH_PT_SETUP") == "2":  # -DSQPH_PT_SETUP=2: sub-phases of the sparse loading and the lane maps
    sub = {3: "load_sparse", 4: "row lane map", 5: "column lane map", 6: "(before placement)", 8: "slot placement"}
else:
    sub = None
names = ["load+maps", "form_S", "jacobi", "A'w", "stages", "x~", "A x~ + upd", "eliminate", "barrier", "total", "misc", "factor tail", "el: wait A", "el: phase A", "el: wait B", "el: phase B"]
per_iter = ("A'w", "stages", "x~", "A x~ + upd", "barrier")
if sub:
    names = [sub.get(i, nm) for i, nm in enumerate(names)]
    per_iter = ()
t = x[:, :16].mean(axis=0)
print(s.kernel_name(), "kernel ms", ms, "ticks of wave 0:")
for nm, val in zip(names, t):
    if nm == "-":
        continue
    per = val / iters if nm in per_iter else val
    print("  %-12s %12.0f %s" % (nm, per, "per iter" if nm in per_iter else ""))
