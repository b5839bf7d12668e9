"""exec_m_transpose / exec_v_map with the unary-op family.

CPU: the oracle's restatement against the genuine reference (its own unit case tests/test_matrix.cpp:190-215 and randomised
inputs over type x unary op).  GPU: the HIP kernels through the kernel-level C ABI and the operator API against the oracle —
bit-exact for every op that is exactly rounded; the transcendental fp32 ops may differ by 1 ulp in < 0.1 % of the entries."""
import ctypes as C

import numpy as np
import pytest

from helpers import NP, random_csr, random_values

EXACT_UNOPS = ["IDENTITY", "AINV", "MINV", "LNOT", "UONE", "ABS"]
FLOAT_EXACT = ["SQRT", "CEIL", "FLOOR", "ROUND", "TRUNC"]
FLOAT_LIBM = ["LOG", "EXP", "SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN"]


def unops_for(t):
    if t == "FLOAT":
        return EXACT_UNOPS + FLOAT_EXACT + FLOAT_LIBM
    return EXACT_UNOPS + ["BNOT"]


def values_for(rng, n, t, op):
    x = random_values(rng, n, t, small=True)
    if op == "MINV":  # the reference's integer 1 / 0 is undefined behaviour: keep clear of it
        x[x == 0] = 1
    if t == "FLOAT":
        if op in ("ASIN", "ACOS"):
            x = (x / 4.0).astype(np.float32).clip(-1, 1)
        if op in ("CEIL", "FLOOR", "ROUND", "TRUNC", "ABS", "AINV"):
            x = (x - 2.03125).astype(np.float32)  # negatives and halves: 0.5 steps exercise round-half-away
    return x


def same(t, op, got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    if t == "FLOAT" and op in FLOAT_LIBM:
        # the reference evaluates these in double and rounds once; so does the device, whose double libm is within 1-2 ulp (of a
        # double) of glibc's: the float results differ only where that moves a value across a rounding boundary — at most 1 ulp,
        # and rarely
        ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        if len(ulp):
            assert ulp.max() <= 1 and (ulp != 0).mean() < 1e-3, (op, ulp.max(), (ulp != 0).mean())
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), op


# ---- CPU: oracle == reference ----------------------------------------------------------------------------------------------
def test_reference_unit_case_transpose(oracle, reference):
    # reference tests/test_matrix.cpp:190-215: 100 x 200 int32, entries where (i + j) odd = i * 10 + j, AINV on the way
    M, N = 100, 200
    I, J = np.nonzero((np.arange(M)[:, None] + np.arange(N)[None, :]) % 2)
    Ap = np.zeros(M + 1, np.uint32)
    np.add.at(Ap, I + 1, 1)
    Ap = np.cumsum(Ap).astype(np.uint32)
    Aj, Ax = J.astype(np.uint32), (I * 10 + J).astype(np.int32)
    for impl in (oracle, reference):
        Rp, Rj, Rx = impl.m_transpose("INT", M, N, Ap, Aj, Ax, "AINV")
        dense = np.zeros((N, M), np.int64)
        rows = np.repeat(np.arange(N), np.diff(Rp.astype(np.int64)))
        dense[rows, Rj] = Rx
        want = np.where((np.arange(N)[:, None] + np.arange(M)[None, :]) % 2, -(np.arange(M)[None, :] * 10 + np.arange(N)[:, None]), 0)
        assert np.array_equal(dense, want), impl.kind


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_oracle_matches_reference_transpose(oracle, reference, t):
    rng = np.random.default_rng(7)
    for op in unops_for(t):
        n_rows, n_cols = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 4, t, skew=True)
        Ax = values_for(rng, len(Aj), t, op)
        a, b = oracle.m_transpose(t, n_rows, n_cols, Ap, Aj, Ax, op), reference.m_transpose(t, n_rows, n_cols, Ap, Aj, Ax, op)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), op
        same(t, "IDENTITY", a[2], b[2])  # both are host libm: bit-exact for every op


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_oracle_matches_reference_map(oracle, reference, t):
    rng = np.random.default_rng(8)
    for op in unops_for(t):
        n = int(rng.integers(1, 200))
        x = values_for(rng, n, t, op)
        same(t, "IDENTITY", oracle.v_map(t, x, op), reference.v_map(t, x, op))
        idx = np.sort(rng.choice(n, size=max(1, n // 3), replace=False)).astype(np.uint32)
        ri, rx = reference.v_map(t, x[idx], op, idx=idx, n=n)
        assert np.array_equal(ri, idx), op  # the index list is copied as stored (cpu_v_map.hpp:89-92)
        same(t, "IDENTITY", oracle.v_map(t, x[idx], op), rx)


# ---- GPU: HIP == oracle ----------------------------------------------------------------------------------------------------
def hip_transpose(t, n_rows, n_cols, Ap, Aj, Ax, op):
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import type_id, unop_id
    nnz = len(Aj)
    dAp, dAj, dAx = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t]))
    Rp = torch.full((n_cols + 1,), -1, dtype=torch.int32, device="cuda")
    Rj = torch.full((max(nnz, 1),), -1, dtype=torch.int32, device="cuda")
    Rx = torch.full((max(nnz, 1),), -1, dtype=torch.int32, device="cuda")
    _lib.call("splahip_m_transpose", type_id(t), n_rows, n_cols, nnz, _ptr(dAp), _ptr(dAj) if nnz else None, _ptr(dAx) if nnz else None,
              _ptr(Rp), _ptr(Rj), _ptr(Rx), unop_id(op), current_stream_ptr())
    torch.cuda.synchronize()
    return host(Rp, np.uint32), host(Rj, np.uint32)[:nnz], host(Rx, np.uint32)[:nnz].view(NP[t])


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_transpose_every_unary_op(oracle, t):
    rng = np.random.default_rng(11)
    for op in unops_for(t):
        n_rows, n_cols = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
        Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 6, t, skew=True, long_rows=2, long_len=700)
        Ax = values_for(rng, len(Aj), t, op)
        want = oracle.m_transpose(t, n_rows, n_cols, Ap, Aj, Ax, op)
        got = hip_transpose(t, n_rows, n_cols, Ap, Aj, Ax, op)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), op
        same(t, op, got[2], want[2])


@pytest.mark.gpu
def test_transpose_edge_shapes(oracle):
    rng = np.random.default_rng(12)
    cases = [(1, 1, 1), (1, 5000, 40), (5000, 1, 1), (257, 63, 0), (4096, 4096, 1), (300, 70000, 9), (70000, 300, 2)]
    for n_rows, n_cols, deg in cases:
        Ap, Aj, _ = random_csr(rng, n_rows, n_cols, min(deg, n_cols), "INT", empty_frac=0.5 if deg else 1.0)
        Ax = random_values(rng, len(Aj), "INT", small=False)
        want = oracle.m_transpose("INT", n_rows, n_cols, Ap, Aj, Ax, "IDENTITY")
        got = hip_transpose("INT", n_rows, n_cols, Ap, Aj, Ax, "IDENTITY")
        for a, b in zip(got, want):
            assert np.array_equal(a, b), (n_rows, n_cols, deg)


@pytest.mark.gpu
def test_transpose_twice_is_identity_at_scale():
    # size-independent property at a size the oracle would take long for: (M^T)^T == M, entry for entry
    from spla_amd import graphgen
    n, Ap, Aj = graphgen.rmat_csr(18, 16, seed=3)
    Ap, Aj = Ap.numpy().view(np.uint32), Aj.numpy().view(np.uint32)
    Ax = (np.arange(len(Aj)) % 1000003).astype(np.int32)
    Tp, Tj, Tx = hip_transpose("INT", n, n, Ap, Aj, Ax, "AINV")
    assert int(Tp[-1]) == len(Aj)
    rows = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Tp.astype(np.int64)))
    assert np.all(np.diff(Tp.astype(np.int64)) >= 0) and np.all((Tj[1:] > Tj[:-1]) | (rows[1:] != rows[:-1]))  # rows ascending inside a row of R
    Bp, Bj, Bx = hip_transpose("INT", n, n, Tp, Tj, Tx, "AINV")
    assert np.array_equal(Bp, Ap) and np.array_equal(Bj, Aj) and np.array_equal(Bx, Ax)


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_map_kernel(oracle, t):
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import type_id, unop_id
    rng = np.random.default_rng(13)
    for op in unops_for(t):
        n = int(rng.integers(1, 100000))
        x = values_for(rng, n, t, op)
        dx = dev(x)
        out = torch.empty_like(dx)
        _lib.call("splahip_map", type_id(t), n, _ptr(dx), _ptr(out), unop_id(op), current_stream_ptr())
        torch.cuda.synchronize()
        same(t, op, host(out, np.uint32).view(NP[t]), oracle.v_map(t, x, op))


@pytest.mark.gpu
def test_operator_api_transpose_and_map(oracle):
    """through spla_Exec_m_transpose / spla_Exec_v_map, in the call shape of pyspla's Matrix.transpose / Vector.map"""
    from spla_amd import api
    rng = np.random.default_rng(14)
    for ty, t in ((api.INT, "INT"), (api.FLOAT, "FLOAT")):
        n_rows, n_cols = 180, 95
        Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 5, t)
        Ax = values_for(rng, len(Aj), t, "AINV")
        M = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), ty)
        R = M.transpose(ty.AINV)
        I, J, V = R.to_lists()
        Rp, Rj, Rx = oracle.m_transpose(t, n_rows, n_cols, Ap, Aj, Ax, "AINV")
        rows = np.repeat(np.arange(n_cols), np.diff(Rp.astype(np.int64)))
        assert np.array_equal(np.asarray(I), rows) and np.array_equal(np.asarray(J), Rj)
        same(t, "AINV", np.asarray(V, NP[t]), Rx)
        # map: dense and sparse
        x = values_for(rng, 500, t, "ABS")
        v = api.Vector.dense(500, ty)
        idx = np.arange(500, dtype=np.uint32)
        v.build(idx, x)
        same(t, "ABS", v.map(ty.ABS).to_dense(), oracle.v_map(t, x, "ABS"))
    # a user-defined op has no device code: NotImplemented, never a host fallback
    assert not hasattr(api.FLOAT, "NOPE")
    with pytest.raises(AttributeError):
        api.FLOAT.BNOT
