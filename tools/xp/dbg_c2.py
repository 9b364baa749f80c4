import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from sqp_solver_amd import QPSolverBatch
from sqp_solver_amd.problems import random_qp_batch_torch
dev = torch.device("cuda", 0)
def run(tag):
    d = random_qp_batch_torch(4096, 20, 40, seed=20250228 + 2, dtype=torch.float64, device=dev)
    s = QPSolverBatch(20, 40, 4096, dtype=np.float64, device=0)
    bench.apply_mode(s.settings, "fixed", 200)
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    P, q, A, l, u = d
    for _ in range(2): s.setup_solve(P, q, A, l, u, colmajor=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); s.setup_solve(P, q, A, l, u, colmajor=True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, ["%.2f" % t for t in ts])
    s.close()
run("fresh")
import oracle
P, q, A, l, u = random_qp_batch_torch(64, 50, 100, seed=1, dtype=torch.float64, device=dev)
oracle.solve_batch(P.cpu().numpy().transpose(0, 2, 1), q.cpu().numpy(), A.cpu().numpy().transpose(0, 2, 1), l.cpu().numpy(), u.cpu().numpy(), settings=oracle.default_settings(), nthreads=oracle.max_threads(), dtype=np.float64)
run("after an oracle call on %d threads" % oracle.max_threads())
time.sleep(1.0)
run("one second later")
