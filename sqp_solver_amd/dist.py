"""Multi-GPU plumbing: the batch shards embarrassingly (one process per GPU, a contiguous block of
QPs each, no data-path collective); the only exchange is the final collection of result records
(x, y, info) on rank 0 — a gather over RCCL/xGMI ("nccl" backend) or gloo in the CPU tests."""
import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous block split of `total` QPs over `world` ranks (first `total % world` ranks get one more)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_views(solver, device):
    """torch tensors aliasing the solver's resident x [B,n], y [B,m], info bytes [B,40]."""
    import torch

    xp, yp, zp, ip = solver.device_state_ptrs()
    ts = "<f8" if solver.dtype == np.float64 else "<f4"
    B = solver.batch
    x = torch.as_tensor(_DevArray(xp, (B, solver.n), ts), device=device)
    y = torch.as_tensor(_DevArray(yp, (B, max(solver.m, 1)), ts), device=device)
    info = torch.as_tensor(_DevArray(ip, (B, 40), "|u1"), device=device)
    return x, y, info


class ResultGather:
    """Collect per-rank result tensors on rank `dst` with one gather per array."""

    def __init__(self, solver=None, world=1, rank=0, device=None, tensors=None, dst=0):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.world, self.rank, self.dst = world, rank, dst
        self.local = list(tensors) if tensors is not None else list(device_views(solver, device))
        self.out = None
        if rank == dst:
            self.out = [[torch.empty_like(t) for _ in range(world)] for t in self.local]

    def gather(self):
        for i, t in enumerate(self.local):
            self.dist.gather(t, self.out[i] if self.rank == self.dst else None, dst=self.dst)
        return self.out

    def stacked(self):
        """On dst: each array concatenated over ranks in rank order (== the unsharded batch order)."""
        import torch

        if self.out is None:
            return None
        return [torch.cat(parts, dim=0) for parts in self.out]
