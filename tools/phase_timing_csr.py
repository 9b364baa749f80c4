"""Debug helper (GPU): build with -DSQPH_PHASE_TIMING and print the per-phase tick breakdown of the sparse kernel."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sqp_solver_amd import build as b
if not os.environ.get("SQPH_LIB"):  # SQPH_LIB = a prebuilt -DSQPH_PHASE_TIMING library (tools/slim_build.sh ... -DSQPH_SLIM_CSR -DSQPH_PHASE_TIMING)
    b.FLAGS.append("-DSQPH_PHASE_TIMING")
    for f in os.environ.get("SQPH_EXTRA_FLAGS", "").split():
        b.FLAGS.append(f)
    b.LIB = b.LIB.replace("libsqp_hip.so", "libsqp_hip_timing.so")
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-o", b.LIB, os.path.join(b.CSRC, "capi.hip"), os.path.join(b.CSRC, "wg_nocheck.hip")])
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
args = [tile(a) for a in (P, q, rp, ci, v, l, u)]
s.setup_solve_csr(*args)
s.enable_timing(True)
s.setup_solve_csr(*args)
ms = s.collect_kernel_ms()
x, y, z, info = s.solution()
names = ["load_sparse", "form_S", "eliminate", "csc", "stage_W", "quad_y1", "stage_WT", "quad_x", "csr+upd", "total", "misc"]
t = x[:, :11].mean(axis=0)
print(s.kernel_name(), "kernel ms", ms, "ticks (100 MHz?):")
for nm, val in zip(names, t):
    per = val / iters if nm in ("csc", "stage_W", "quad_y1", "stage_WT", "quad_x", "csr+upd") else val
    print("  %-12s %12.0f %s" % (nm, per, "per iter" if per != val else ""))
