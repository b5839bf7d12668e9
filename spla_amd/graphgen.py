"""Deterministic synthetic inputs for the hot path (the reference ships no generator).

* ``rmat_csr``    — Graph500-style R-MAT (a,b,c,d) = (0.57,0.19,0.19,0.05), edge factor 16, self loops and duplicate
                    edges removed, directed (no symmetrisation), optional random vertex relabelling.
* ``uniform_csr`` — n x n with nnz distinct uniformly drawn (i, j) pairs (BASELINE config 1).

Generation runs in torch on whichever device is asked for (the GPU for the bench sizes, the CPU for tests) with a
seeded generator; the result is a CSR triple of int32 tensors (uint32 words) on that device.
"""
from __future__ import annotations

import torch


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def owner_sorted_permutation(n: int, owners: int, seed: int, device="cpu") -> torch.Tensor:
    """old label -> new label such that every one of `owners` equal blocks of the new id space holds a pseudo-random 1/owners of the
    vertices IN THEIR ORIGINAL ORDER.  A 1-D row-block split over these labels is balanced like a random relabelling (every block
    gets its share of the hubs), but inside a block low raw ids — R-MAT's hot vertices — stay clustered at the front, so the
    gathered vector keeps one hot region per block instead of hot entries scattered over every cache line."""
    owner = torch.randperm(n, generator=_gen(device, seed + 7919), device=device) % owners
    order = torch.argsort(owner * n + torch.arange(n, device=device))  # vertices grouped by owner, ascending raw id inside a group
    perm = torch.empty(n, dtype=torch.int64, device=device)
    perm[order] = torch.arange(n, device=device)
    return perm


def rmat_edge_keys(scale: int, edge_factor: int = 16, a: float = 0.57, b: float = 0.19, c: float = 0.19,
                   seed: int = 1, device="cpu", chunk: int = 1 << 25, permute=False, owners: int = 0) -> torch.Tensor:
    """Sorted unique int64 keys i * n + j of the R-MAT edges (self loops removed).  permute: False = raw labels, True = random
    relabelling, "owner-sorted" = owner_sorted_permutation(n, owners)."""
    n = 1 << scale
    m = edge_factor << scale
    g = _gen(device, seed)
    ab, abc = a + b, a + b + c
    perm = None
    if permute == "owner-sorted":
        perm = owner_sorted_permutation(n, max(1, owners), seed, device)
    elif permute:
        perm = torch.randperm(n, generator=_gen(device, seed + 7919), device=device)
    keys = []
    done = 0
    while done < m:
        cnt = min(chunk, m - done)
        i = torch.zeros(cnt, dtype=torch.int64, device=device)
        j = torch.zeros(cnt, dtype=torch.int64, device=device)
        for _ in range(scale):
            u = torch.rand(cnt, generator=g, device=device)
            row_bit = u >= ab
            col_bit = ((u >= a) & (u < ab)) | (u >= abc)
            i = (i << 1) | row_bit.to(torch.int64)
            j = (j << 1) | col_bit.to(torch.int64)
        if perm is not None:
            i, j = perm[i], perm[j]
        keep = i != j
        keys.append(i[keep] * n + j[keep])
        done += cnt
        del i, j, u, row_bit, col_bit, keep
    key = torch.cat(keys) if len(keys) > 1 else keys[0]
    del keys
    return torch.unique(key, sorted=True)


def keys_to_csr(key: torch.Tensor, n_rows: int, n_cols: int):
    """(Ap, Aj) int32 tensors from sorted unique keys i * n_cols + j."""
    rows = torch.div(key, n_cols, rounding_mode="floor")
    Aj = (key - rows * n_cols).to(torch.int32)
    counts = torch.bincount(rows, minlength=n_rows)
    Ap = torch.zeros(n_rows + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(counts, 0, out=Ap[1:])
    assert int(Ap[-1]) < 2 ** 31, "nnz must fit uint32 indices (reference include/spla/config.hpp:56)"
    return Ap.to(torch.int32), Aj


def rmat_csr(scale: int, edge_factor: int = 16, seed: int = 1, device="cpu", permute=False, owners: int = 0):
    """(n, Ap, Aj) of the R-MAT graph; values are attached by the caller."""
    n = 1 << scale
    key = rmat_edge_keys(scale, edge_factor, seed=seed, device=device, permute=permute, owners=owners)
    Ap, Aj = keys_to_csr(key, n, n)
    return n, Ap, Aj


def uniform_csr(n: int, nnz: int, seed: int = 1, device="cpu"):
    """(Ap, Aj) with exactly nnz distinct uniformly drawn entries (BASELINE config 1: n=100k, nnz=1M)."""
    g = _gen(device, seed)
    key = torch.empty(0, dtype=torch.int64, device=device)
    while key.numel() < nnz:
        need = nnz - key.numel()
        draw = torch.randint(0, n * n, (int(need * 1.1) + 16,), generator=g, device=device, dtype=torch.int64)
        key = torch.unique(torch.cat([key, draw]), sorted=True)
    if key.numel() > nnz:
        # drop a deterministic random subset to hit nnz exactly
        drop = torch.randperm(key.numel(), generator=g, device=device)[: key.numel() - nnz]
        keep = torch.ones(key.numel(), dtype=torch.bool, device=device)
        keep[drop] = False
        key = key[keep]
    return keys_to_csr(key, n, n)


def degree_stats(Ap: torch.Tensor) -> dict:
    """min / avg / sd / max of the row lengths, as the reference's MtxLoader prints (src/io.cpp:273-295)."""
    deg = (Ap[1:] - Ap[:-1]).to(torch.float64)
    return {"min": float(deg.min()), "avg": float(deg.mean()), "sd": float(deg.std(unbiased=False)),
            "max": float(deg.max()), "empty_rows": int((deg == 0).sum())}


def nnz_balanced_row_splits(Ap: torch.Tensor, parts: int):
    """1-D row-block boundaries (parts + 1 row indices) that balance nnz, from the prefix array Ap."""
    nnz = int(Ap[-1])
    n_rows = Ap.numel() - 1
    targets = torch.tensor([nnz * p // parts for p in range(1, parts)], dtype=Ap.dtype, device=Ap.device)
    cuts = torch.searchsorted(Ap.contiguous(), targets, right=False).clamp_(0, n_rows)
    bounds = [0] + [int(x) for x in cuts.tolist()] + [n_rows]
    for k in range(1, len(bounds)):
        bounds[k] = max(bounds[k], bounds[k - 1])
    return bounds


def chung_lu_csr(n: int, nnz_target: int, max_degree: int, seed: int = 1, device="cpu", chunk: int = 1 << 25):
    """(n, Ap, Aj) of a SYMMETRIC Chung-Lu graph with a power-law expected degree sequence, matched to three numbers of a real graph: vertices n, stored
    entries nnz_target (both directions of every edge), largest degree max_degree.  Expected degrees w_i = c (i + 1)^-alpha with alpha solved from
    max / mean = (1 - alpha) n^alpha; both endpoints of an edge are drawn proportionally to w (inverse-CDF sampling), self loops and duplicates dropped,
    vertex labels randomly permuted (a real graph's labels carry no degree order).  bench_workloads.py uses it as the stand-in for SuiteSparse com-Orkut
    (n = 3,072,441, 234.4 M entries, max degree 33,313: reference README.md:148), which is not on the box."""
    import math
    mean = nnz_target / n
    ratio = max_degree / mean
    lo, hi = 0.01, 0.99
    for _ in range(60):  # (1 - a) n^a is increasing in a on this interval for n >> 1
        mid = 0.5 * (lo + hi)
        if (1.0 - mid) * n ** mid < ratio:
            lo = mid
        else:
            hi = mid
    alpha = 0.5 * (lo + hi)
    g = _gen(device, seed)
    w = (torch.arange(1, n + 1, device=device, dtype=torch.float64)) ** (-alpha)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    perm = torch.randperm(n, generator=_gen(device, seed + 7919), device=device)
    m = int(nnz_target // 2 * 1.002)  # undirected edges to draw (a few per mille are duplicates / self loops)
    keys = []
    done = 0
    while done < m:
        cnt = min(chunk, m - done)
        u = torch.searchsorted(cdf, torch.rand(cnt, generator=g, device=device, dtype=torch.float64)).clamp_(max=n - 1)
        v = torch.searchsorted(cdf, torch.rand(cnt, generator=g, device=device, dtype=torch.float64)).clamp_(max=n - 1)
        u, v = perm[u], perm[v]
        keep = u != v
        u, v = u[keep], v[keep]
        keys.append(u * n + v)
        keys.append(v * n + u)
        done += cnt
        del u, v, keep
    key = torch.unique(torch.cat(keys), sorted=True)
    del keys
    Ap, Aj = keys_to_csr(key, n, n)
    return n, Ap, Aj
