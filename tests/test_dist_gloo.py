"""N>1 path on CPU: two processes (gloo), each solving its contiguous shard of a batch with the emulated
HIP kernel, results collected on rank 0 with the same ResultGather code bench.py uses over RCCL."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import simlib
from sqp_solver_amd.dist import shard_bounds, ResultGather
from sqp_solver_amd.problems import random_qp_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, m, total = 6, 9, 7
P, q, A, l, u = random_qp_batch(total, n, m, seed=77)
lo, hi = shard_bounds(total, world, rank)
s = simlib.SimSolverBatch(n, m, hi - lo, variant=simlib.WG)
s.settings.max_iter = 40; s.settings.check_termination = 0
s.setup_solve(P[lo:hi], q[lo:hi], A[lo:hi], l[lo:hi], u[lo:hi])
x, y, z, info = s.solution()
# unequal shards (total = 7 over 2 ranks: 4 + 3): ResultGather pads the staging buffers itself
tx, ty, ti = torch.from_numpy(x.copy()), torch.from_numpy(y.copy()), torch.from_numpy(info.iter.astype(np.int32).reshape(-1, 1))
g = ResultGather(world=world, rank=rank, tensors=[tx, ty, ti])
assert g.rows == [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
# three batches through the two staging buffers (the third reuses the first one after its gather has completed);
# the arrays change in between, as the solver state does from batch to batch
keep = tx.clone()
tx.mul_(0.5); g.gather()
tx.mul_(3.0); g.gather()
tx.copy_(keep); g.gather()
if rank == 0:
    xs, ys, its = [t.numpy() for t in g.stacked()]
    assert xs.shape[0] == total
    np.savez(sys.argv[2], x=xs, y=ys, iter=its[:, 0])
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_shard_and_gather(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import simlib
    from sqp_solver_amd.problems import random_qp_batch

    simlib.lib()  # build the emulator library once here: the two ranks would otherwise race in `make`
    out = str(tmp_path / "gathered.npz")
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", "29533", script, ROOT, out], env=env, timeout=600)
    g = np.load(out)
    n, m, total = 6, 9, 7
    P, q, A, l, u = random_qp_batch(total, n, m, seed=77)
    s = simlib.SimSolverBatch(n, m, total, variant=simlib.WG)
    s.settings.max_iter = 40
    s.settings.check_termination = 0
    s.setup_solve(P, q, A, l, u)
    x, y, z, info = s.solution()
    assert np.array_equal(g["x"], x) and np.array_equal(g["y"], y) and np.array_equal(g["iter"], info.iter)
