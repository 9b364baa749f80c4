"""Deterministic synthetic QP batches.

Per QP:  G ~ N(0,1)^{n x n},  P = G G'/n + 0.1 I  (SPD, exactly symmetric);  q ~ N(0,1);
A ~ N(0,1)^{m x n};  x0 ~ N(0,1), c = A x0, l = c - U(0,1), u = c + U(0,1); then by a row hash
10 % of the rows become equalities (u = l = c), 10 % one-sided (u = +inf) and 2 % loose
(l = -1e20, u = +1e20), so constraint classification and the three rho classes are exercised.
"""
import numpy as np


def _row_classes(batch, m):
    b = np.arange(batch, dtype=np.uint64)[:, None]
    i = np.arange(m, dtype=np.uint64)[None, :]
    h = (b * np.uint64(2654435761) + i * np.uint64(40503) + np.uint64(12345)) % np.uint64(100)
    return h.astype(np.int64)


def random_qp_batch(batch, n, m, seed=20250228, dtype=np.float64, plain=False):
    """Returns P [B,n,n], q [B,n], A [B,m,n], l [B,m], u [B,m] (math indexing)."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((batch, n, n))
    P = G @ np.transpose(G, (0, 2, 1)) / n + 0.1 * np.eye(n)[None]
    P = 0.5 * (P + np.transpose(P, (0, 2, 1)))
    q = rng.standard_normal((batch, n))
    A = rng.standard_normal((batch, m, n))
    x0 = rng.standard_normal((batch, n))
    c = np.einsum("bij,bj->bi", A, x0)
    l = c - rng.uniform(0, 1, (batch, m))
    u = c + rng.uniform(0, 1, (batch, m))
    if not plain and m > 0:
        h = _row_classes(batch, m)
        eq = h < 10
        one = (h >= 10) & (h < 20)
        loose = (h >= 20) & (h < 22)
        l = np.where(eq, c, l)
        u = np.where(eq, c, u)
        u = np.where(one, np.inf, u)
        l = np.where(loose, -1e20, l)
        u = np.where(loose, 1e20, u)
    cast = lambda a: np.ascontiguousarray(a, dtype=dtype)  # noqa: E731
    return cast(P), cast(q), cast(A), cast(l), cast(u)


def random_qp_batch_torch(batch, n, m, seed=20250228, dtype=None, device="cuda", chunk=4096):
    """Same distribution generated on the GPU, directly in the C-ABI's per-QP column-major layout.

    Returns P_cm [B,n,n] (symmetric), q [B,n], A_cm [B,n,m] with A_cm[b,j,i] = A[b,i,j], l, u.
    """
    import torch

    dtype = dtype or torch.float64
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    P = torch.empty((batch, n, n), dtype=dtype, device=device)
    q = torch.empty((batch, n), dtype=dtype, device=device)
    A_cm = torch.empty((batch, n, m), dtype=dtype, device=device)
    l = torch.empty((batch, m), dtype=dtype, device=device)
    u = torch.empty((batch, m), dtype=dtype, device=device)
    eye = torch.eye(n, dtype=dtype, device=device)
    hb = torch.arange(batch, device=device, dtype=torch.int64)
    hi = torch.arange(m, device=device, dtype=torch.int64)
    for s in range(0, batch, chunk):
        e = min(batch, s + chunk)
        B = e - s
        G = torch.randn((B, n, n), generator=gen, dtype=dtype, device=device)
        Pc = G @ G.transpose(1, 2) / n + 0.1 * eye
        P[s:e] = 0.5 * (Pc + Pc.transpose(1, 2))
        q[s:e] = torch.randn((B, n), generator=gen, dtype=dtype, device=device)
        At = torch.randn((B, n, m), generator=gen, dtype=dtype, device=device)  # A' per QP
        A_cm[s:e] = At
        x0 = torch.randn((B, n), generator=gen, dtype=dtype, device=device)
        c = torch.einsum("bji,bj->bi", At, x0)
        lo = c - torch.rand((B, m), generator=gen, dtype=dtype, device=device)
        hi_ = c + torch.rand((B, m), generator=gen, dtype=dtype, device=device)
        h = (hb[s:e, None] * 2654435761 + hi[None, :] * 40503 + 12345) % 100
        eq = h < 10
        one = (h >= 10) & (h < 20)
        loose = (h >= 20) & (h < 22)
        lo = torch.where(eq, c, lo)
        hi_ = torch.where(eq, c, hi_)
        hi_ = torch.where(one, torch.full_like(hi_, float("inf")), hi_)
        lo = torch.where(loose, torch.full_like(lo, -1e20), lo)
        hi_ = torch.where(loose, torch.full_like(hi_, 1e20), hi_)
        l[s:e] = lo
        u[s:e] = hi_
    return P, q, A_cm, l, u


def random_csr_qp_batch(batch, n, m, density=0.05, seed=20250233, dtype=np.float64, shared_pattern=False):
    """BASELINE config 5: the same distribution with A sparse — `density` of the entries kept, at least
    one per row.  Returns P [B,n,n], q, rowptr int32 [B,m+1], colind int32 [B,nnz_max], val [B,nnz_max] (rows of each
    QP stored back to back, zero-padded to nnz_max), l, u and the dense A [B,m,n] the CSR arrays encode."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((batch, n, n))
    P = G @ np.transpose(G, (0, 2, 1)) / n + 0.1 * np.eye(n)[None]
    P = 0.5 * (P + np.transpose(P, (0, 2, 1)))
    q = rng.standard_normal((batch, n))
    pb = 1 if shared_pattern else batch
    mask = rng.uniform(0, 1, (pb, m, n)) < density
    forced = rng.integers(0, n, (pb, m))
    mask[np.arange(pb)[:, None], np.arange(m)[None, :], forced] = True
    if shared_pattern:
        mask = np.broadcast_to(mask, (batch, m, n))
    A = rng.standard_normal((batch, m, n)) * mask
    x0 = rng.standard_normal((batch, n))
    c = np.einsum("bij,bj->bi", A, x0)
    l = c - rng.uniform(0, 1, (batch, m))
    u = c + rng.uniform(0, 1, (batch, m))
    h = _row_classes(batch, m)
    eq, one, loose = h < 10, (h >= 10) & (h < 20), (h >= 20) & (h < 22)
    l = np.where(eq, c, l)
    u = np.where(eq, c, u)
    u = np.where(one, np.inf, u)
    l = np.where(loose, -1e20, l)
    u = np.where(loose, 1e20, u)
    counts = mask.sum(axis=2)
    rowptr = np.zeros((batch, m + 1), dtype=np.int32)
    rowptr[:, 1:] = np.cumsum(counts, axis=1)
    nnz_max = int(rowptr[:, -1].max())
    colind = np.zeros((batch, nnz_max), dtype=np.int32)
    val = np.zeros((batch, nnz_max), dtype=dtype)
    for b in range(batch):
        ii, jj = np.nonzero(mask[b])  # row-major order == CSR order
        colind[b, : len(jj)] = jj
        val[b, : len(jj)] = A[b, ii, jj]
    cast = lambda a: np.ascontiguousarray(a, dtype=dtype)  # noqa: E731
    return cast(P), cast(q), rowptr, colind, val, cast(l), cast(u), cast(A)


SIMPLE_QP = dict(  # the reference's canonical fixture, tests/qp_solver_test.cpp:19-31
    P=np.array([[4.0, 1.0], [1.0, 2.0]]), q=np.array([1.0, 1.0]),
    A=np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]]), l=np.array([1.0, 0.0, 0.0]), u=np.array([1.0, 0.7, 0.7]),
    solution=np.array([0.3, 0.7]), dual=np.array([-2.9, 0.0, 0.2]),
)
