"""Debug helper (GPU): build with -DSQPH_PHASE_TIMING and print the per-phase tick breakdown of the wg kernel."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sqp_solver_amd import build as b
if not os.environ.get("SQPH_LIB"):  # SQPH_LIB=<prebuilt -DSQPH_PHASE_TIMING library> (tools/slim_build.sh) skips the compile
    b.FLAGS.append("-DSQPH_PHASE_TIMING")
    b.LIB = b.LIB.replace("libsqp_hip.so", "libsqp_hip_timing.so")
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-o", b.LIB, os.path.join(b.CSRC, "capi.hip"), os.path.join(b.CSRC, "wg_nocheck.hip")])
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch
n, m, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P, q, A, l, u = random_qp_batch(64, n, m, seed=1)
rep = (B + 63) // 64
tile = lambda a: np.concatenate([a] * rep)[:B]
s = QPSolverBatch(n, m, B)
s.settings.max_iter = 200
s.settings.check_termination = 0
args = [tile(a) for a in (P, q, A, l, u)]
s.setup_solve(*args)
s.setup_solve(*args)
x, y, z, info = s.solution()
names = ["bar1", "stage1", "bar2", "reduce_y1", "bar3", "stage2", "bar4", "update"]
for nm, arr in (("wave0", x[:, :8]), ("wave1", y[:, 64:72])):
    if arr.shape[1] < 8: continue
    t = arr.mean(axis=0) / 200.0
    print(s.kernel_name(), nm, "ticks/iter:", " ".join("%s=%.0f" % (a, v) for a, v in zip(names, t)), "total=%.0f" % t.sum())
