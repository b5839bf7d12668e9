"""Shared input generators for the parity tests (seeded, deterministic)."""
from __future__ import annotations

import numpy as np

NP = {"INT": np.int32, "UINT": np.uint32, "FLOAT": np.float32}

# (op_mult, op_add) pairs exercised per type; bitwise ops are I/U only (reference src/op.cpp:236-241)
SEMIRINGS_COMMON = [("MULT", "PLUS"), ("PLUS", "MIN"), ("LAND", "LOR"), ("MIN", "MAX"), ("MAX", "MIN"),
                    ("PLUS", "MAX"), ("SECOND", "PLUS"), ("FIRST", "MIN"), ("MULT", "MULT"), ("BONE", "PLUS"),
                    ("MULT", "LOR"), ("PLUS", "LAND")]
SEMIRINGS_INT = [("BAND", "BOR"), ("BOR", "BAND"), ("BAND", "BXOR"), ("MULT", "BXOR")]
SEMIRINGS_ORDERED = [("MULT", "FIRST"), ("MULT", "SECOND"), ("MULT", "MINUS"), ("PLUS", "MINUS_POW2"), ("MULT", "BONE")]
SELECTS = ["EQZERO", "NQZERO", "GTZERO", "GEZERO", "LTZERO", "LEZERO", "ALWAYS", "NEVER"]


def random_csr(rng, n_rows, n_cols, avg_deg, t, skew=False, empty_frac=0.2, long_rows=0, long_len=0,
               small_values=True):
    """Random CSR with sorted, duplicate-free columns per row.  Values are small so int products/sums never
    overflow (int32 overflow is UB in the reference) and fp32 sums stay well conditioned."""
    if skew:
        deg = np.minimum((rng.pareto(1.2, n_rows) * avg_deg * 0.4).astype(np.int64), n_cols)
    else:
        deg = rng.poisson(avg_deg, n_rows).astype(np.int64)
    deg = np.minimum(deg, n_cols)
    deg[rng.random(n_rows) < empty_frac] = 0
    for _ in range(long_rows):
        deg[rng.integers(0, n_rows)] = min(long_len, n_cols)
    Ap = np.zeros(n_rows + 1, dtype=np.uint32)
    Ap[1:] = np.cumsum(deg)
    nnz = int(Ap[-1])
    Aj = np.empty(nnz, dtype=np.uint32)
    for i in range(n_rows):
        d = int(deg[i])
        if d:
            if d * 4 > n_cols:
                cols = rng.permutation(n_cols)[:d]
            else:
                cols = np.unique(rng.integers(0, n_cols, size=d * 2 + 8))
                while len(cols) < d:
                    cols = np.unique(np.concatenate([cols, rng.integers(0, n_cols, size=d)]))
                cols = rng.permutation(cols)[:d]
            Aj[Ap[i]:Ap[i + 1]] = np.sort(cols)
    Ax = random_values(rng, nnz, t, small_values)
    return Ap, Aj, Ax


def random_values(rng, n, t, small=True, zeros=0.0):
    if t == "FLOAT":
        x = (rng.integers(1, 64, n) / 16.0).astype(np.float32) if small else rng.random(n).astype(np.float32)
        if not small:
            x += 0.25
    elif t == "INT":
        x = rng.integers(-3 if small else -50, 4 if small else 50, n).astype(np.int32)
    else:
        x = rng.integers(0, 4 if small else 50, n).astype(np.uint32)
    if zeros > 0:
        x[rng.random(n) < zeros] = 0
    return x


def random_mask(rng, n, t):
    if t == "FLOAT":
        return rng.integers(-1, 2, n).astype(np.float32)
    if t == "INT":
        return rng.integers(-1, 2, n).astype(np.int32)
    return rng.integers(0, 2, n).astype(np.uint32)


def assert_same(got, want, t, rtol=1e-5, what=""):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    if t == "FLOAT":
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-6, err_msg=what)
    else:
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=what)
