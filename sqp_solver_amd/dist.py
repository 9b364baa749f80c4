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
    """torch tensors aliasing the solver's resident x [B,n], y [B,m], info bytes [B,40].

    The resident state is fp64 whatever the solver's interface dtype (sqph_device_state, include/sqp_hip.h)."""
    import torch

    xp, yp, zp, ip = solver.device_state_ptrs()
    ts = "<f8"
    B = solver.batch
    x = torch.as_tensor(_DevArray(xp, (B, solver.n), ts), device=device)
    y = torch.as_tensor(_DevArray(yp, (B, max(solver.m, 1)), ts), device=device)
    info = torch.as_tensor(_DevArray(ip, (B, 40), "|u1"), device=device)
    return x, y, info


class ResultGather:
    """Collect the per-rank result records on rank `dst`: the arrays are packed into ONE staging buffer and sent with
    ONE gather per batch (direct peer-to-root over xGMI with the "nccl" backend).  The collective is asynchronous and
    double-buffered: the gather of batch k runs on the communication stream while batch k+1 is being solved; a staging
    buffer is reused only after its previous gather has completed (stream dependency, no host wait).  `flush()` joins
    everything outstanding (call it before the final synchronisation / before reading `stacked()`).

    Shards need not be equal (shard_bounds gives the first `total % world` ranks one QP more): the row counts are exchanged
    once at construction, every rank's staging buffer is padded to the largest shard's record size (a gather moves equal
    sized buffers), and `stacked()` decodes rank r's buffer with rank r's own row count."""

    def __init__(self, solver=None, world=1, rank=0, device=None, tensors=None, dst=0, depth=2):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.world, self.rank, self.dst, self.depth = world, rank, dst, depth
        self.local = list(tensors) if tensors is not None else list(device_views(solver, device))
        self.local = [t if t.is_contiguous() else t.contiguous() for t in self.local]
        rows = self.local[0].shape[0]
        assert all(t.shape[0] == rows for t in self.local), "every array carries one record per QP"
        # bytes of one QP's record in each array (the arrays may be empty on a rank whose shard is empty)
        self.row_bytes = [int(np.prod(t.shape[1:], dtype=np.int64)) * t.element_size() for t in self.local]
        self.nbytes = [rows * rb for rb in self.row_bytes]
        dev = self.local[0].device
        # the gloo backend moves host memory: device-resident arrays are staged through host buffers there (the copy into the staging
        # buffer is the D2H), the "nccl" (RCCL) backend gathers device buffers directly
        if world > 1 and dev.type == "cuda" and dist.get_backend() == "gloo":
            dev = torch.device("cpu")
        counts = torch.tensor([rows], dtype=torch.int64, device=dev)
        if world > 1:
            allc = [torch.zeros_like(counts) for _ in range(world)]
            dist.all_gather(allc, counts)
            self.rows = [int(c.item()) for c in allc]
        else:
            self.rows = [rows]
        total = max(self.rows) * sum(self.row_bytes)
        self.stage = [torch.zeros(total, dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.out = None
        if rank == dst:
            self.out = [[torch.empty(total, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(depth)]
        self.work = [None] * depth
        self.k = 0
        self.last = None

    def gather(self):
        """Pack the current contents of the local arrays and start their gather (returns immediately)."""
        import torch

        slot = self.k % self.depth
        self.k += 1
        if self.work[slot] is not None:
            self.work[slot].wait()  # the staging buffer is free again once its previous gather is done
        off = 0
        for t, nb in zip(self.local, self.nbytes):
            if nb:
                self.stage[slot][off:off + nb].copy_(t.view(torch.uint8).reshape(-1))
            off += nb
        self.work[slot] = self.dist.gather(self.stage[slot], self.out[slot] if self.rank == self.dst else None, dst=self.dst,
                                           async_op=True)
        self.last = slot
        return self.work[slot]

    def flush(self):
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None

    def stacked(self):
        """On dst: each array of the most recent gather concatenated over ranks in rank order (== the unsharded batch
        order, padding removed); joins outstanding gathers first."""
        import torch

        self.flush()
        if self.out is None or self.last is None:
            return None
        res = []
        for k, t in enumerate(self.local):
            parts = []
            for r, o in enumerate(self.out[self.last]):
                off = self.rows[r] * sum(self.row_bytes[:k])  # rank r packed its arrays back to back with ITS row count
                nb = self.rows[r] * self.row_bytes[k]
                parts.append(o[off:off + nb].view(t.dtype).reshape((self.rows[r],) + tuple(t.shape[1:])))
            res.append(torch.cat(parts, dim=0))
        return res
