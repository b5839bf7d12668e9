"""Pins the oracle (oracle/oracle.c) against the GENUINE reference CPU backend (oracle/_ref/libspla_ref.so):
the reference's own unit-test cases and randomised inputs over every (type x semiring x select x early-exit) cell.
CPU only."""
import numpy as np
import pytest

from helpers import (NP, SELECTS, SEMIRINGS_COMMON, SEMIRINGS_INT, SEMIRINGS_ORDERED, assert_same, random_csr,
                     random_mask, random_values)


def test_reference_unit_case_mxv_naive(oracle, reference):
    # reference tests/test_mxv.cpp:33-89 : 4x5 int32, MULT/PLUS, EQZERO mask, init 0 -> [0,14,0,1]
    Ap = np.array([0, 2, 4, 5, 6], np.uint32)
    Aj = np.array([1, 4, 0, 4, 2, 4], np.uint32)
    Ax = np.array([2, -9, 2, -8, 3, -1], np.int32)
    mask = np.array([1, 0, 1, 0], np.int32)
    v = np.array([3, 0, 3, 0, -1], np.int32)
    for impl in (oracle, reference):
        r = impl.mxv_masked("INT", 4, 5, Ap, Aj, Ax, mask, v, "MULT", "PLUS", "EQZERO", 0)
        assert r.tolist() == [0, 14, 0, 1], impl.kind


def test_reference_probes(oracle, reference):
    # SURVEY.md 8c probes, verified against the reference build
    # (iii) fp32 PLUS is a strict left fold: [1e8, 1, -1e8] . 1 -> 0 exactly
    Ap = np.array([0, 3], np.uint32); Aj = np.array([0, 1, 2], np.uint32)
    Ax = np.array([1e8, 1, -1e8], np.float32)
    for impl in (oracle, reference):
        r = impl.mxv_masked("FLOAT", 1, 3, Ap, Aj, Ax, np.ones(1, np.float32), np.ones(3, np.float32), "MULT", "PLUS", "ALWAYS", 0.0)
        assert r[0] == 0.0
    # int32 min-plus with INF = 2^30
    INF = 1 << 30
    Ap = np.array([0, 2, 3, 4], np.uint32); Aj = np.array([1, 2, 2, 0], np.uint32); Ax = np.array([5, 9, 3, 1], np.int32)
    for impl in (oracle, reference):
        r = impl.mxv_masked("INT", 3, 3, Ap, Aj, Ax, np.zeros(3, np.int32), np.array([0, INF, INF], np.int32), "PLUS", "MIN", "ALWAYS", INF)
        assert r.tolist() == [INF, INF, 1]
    # (iv) vxm keeps explicit zero products: v=[3,2], M=[[0,5],[.,7]] -> ([0,1],[0,29])
    Ap = np.array([0, 2, 3], np.uint32); Aj = np.array([0, 1, 1], np.uint32); Ax = np.array([0, 5, 7], np.int32)
    for impl in (oracle, reference):
        ri, rx = impl.vxm_masked("INT", 2, 2, Ap, Aj, Ax, np.zeros(2, np.int32), np.array([0, 1], np.uint32), np.array([3, 2], np.int32), "MULT", "PLUS", "ALWAYS")
        assert ri.tolist() == [0, 1] and rx.tolist() == [0, 29]
    # (v) mxm [1,-1] . [4,4]^T -> empty (cancelled entry dropped)
    A = (np.array([0, 2], np.uint32), np.array([0, 1], np.uint32), np.array([1, -1], np.float32))
    B = (np.array([0, 1, 2], np.uint32), np.array([0, 0], np.uint32), np.array([4, 4], np.float32))
    for impl in (oracle, reference):
        Rp, Rj, Rx = impl.mxm("FLOAT", 1, 1, A, B, "MULT", "PLUS", 0.0)
        assert Rp.tolist() == [0, 0] and len(Rj) == 0
    # (vi) mxmT(A,A,mask=ones,GTZERO) on A=[[1,2],[0,3]] -> [[5,6],[6,9]]
    A = (np.array([0, 2, 4], np.uint32), np.array([0, 1, 0, 1], np.uint32), np.array([1, 2, 0, 3], np.int32))
    Mk = (np.array([0, 2, 4], np.uint32), np.array([0, 1, 0, 1], np.uint32), np.array([1, 1, 1, 1], np.int32))
    for impl in (oracle, reference):
        Rp, Rj, Rx = impl.mxmT_masked("INT", 2, Mk, A, A, "MULT", "PLUS", "GTZERO", 0)
        assert Rp.tolist() == [0, 2, 4] and Rj.tolist() == [0, 1, 0, 1] and Rx.tolist() == [5, 6, 6, 9]


def _semirings(t):
    s = SEMIRINGS_COMMON + SEMIRINGS_ORDERED
    if t != "FLOAT":
        s = s + SEMIRINGS_INT
    return s


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxv_oracle_equals_reference_all_cells(oracle, reference, t):
    rng = np.random.default_rng(7)
    n_rows, n_cols = 97, 83
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 6, t, skew=True)
    for k, (mul, add) in enumerate(_semirings(t)):
        for early in (False, True):
            sel = SELECTS[(k + early) % len(SELECTS)]
            mask = random_mask(rng, n_rows, t)
            v = random_values(rng, n_cols, t, zeros=0.3)
            init = {"MIN": 100, "MAX": 0, "MULT": 1, "LAND": 1}.get(add, 0)
            a = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, early)
            b = reference.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, early)
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=f"{t} {mul}/{add} {sel} early={early}")


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_vxm_oracle_equals_reference(oracle, reference, t):
    rng = np.random.default_rng(11)
    n_rows, n_cols = 71, 64
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 5, t, skew=True)
    for k, (mul, add) in enumerate(_semirings(t)):
        sel = SELECTS[k % len(SELECTS)]
        mask = random_mask(rng, n_cols, t)
        nv = int(rng.integers(0, n_rows))
        vi = np.sort(rng.permutation(n_rows)[:nv]).astype(np.uint32)
        if k % 3 == 0:
            vi = rng.permutation(vi).astype(np.uint32)  # caller-ordered COO: fold order = stored order
        vx = random_values(rng, nv, t, zeros=0.2)
        a = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
        b = reference.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
        np.testing.assert_array_equal(a[0], b[0], err_msg=f"{t} {mul}/{add} {sel} idx")
        np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32), err_msg=f"{t} {mul}/{add} {sel} val")


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxm_and_mxmT_oracle_equals_reference(oracle, reference, t):
    rng = np.random.default_rng(13)
    m, k, n = 37, 41, 29
    A = random_csr(rng, m, k, 4, t)
    B = random_csr(rng, k, n, 4, t)
    for mul, add, init in [("MULT", "PLUS", 0), ("PLUS", "MIN", 100), ("MULT", "SECOND", 0), ("MIN", "MAX", 0)]:
        a = oracle.mxm(t, m, n, A, B, mul, add, init)
        b = reference.mxm(t, m, n, A, B, mul, add, init)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32), err_msg=f"mxm {t} {mul}/{add}")
    Bt = random_csr(rng, n, k, 4, t)
    Mk = random_csr(rng, m, n, 5, t)
    Mk = (Mk[0], Mk[1], random_mask(rng, len(Mk[1]), t))
    for (mul, add, init), sel in zip([("MULT", "PLUS", 0), ("PLUS", "MIN", 100), ("MULT", "SECOND", 0), ("MIN", "MAX", 0)],
                                     ["GTZERO", "ALWAYS", "EQZERO", "NQZERO"]):
        a = oracle.mxmT_masked(t, m, Mk, A, Bt, mul, add, sel, init)
        b = reference.mxmT_masked(t, m, Mk, A, Bt, mul, add, sel, init, n_cols=n, inner=k)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32), err_msg=f"mxmT {t} {mul}/{add} {sel}")


@pytest.mark.parametrize("add", ["MIN", "MAX"])
def test_fp32_min_max_nan_rules_oracle_equals_reference(oracle, reference, add):
    """The NaN rules the HIP kernels are held to (tests/test_*_gpu.py: *nan*), pinned on the reference itself: std::min / std::max never pick a NaN
    second argument, a NaN first argument (init, or the first product of a vxm column) sticks (src/op.cpp:151-155,222-227)."""
    rng = np.random.default_rng(5 + (add == "MAX"))
    t = "FLOAT"
    n_rows, n_cols = 61, 53
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 7, t, skew=True)
    Ax = (rng.integers(-40, 400, len(Aj)) / 8.0).astype(np.float32)
    v = (rng.integers(-500, 5000, n_cols) / 4.0).astype(np.float32)
    v[rng.random(n_cols) < 0.25] = np.nan
    v[rng.random(n_cols) < 0.2] = np.inf
    for init in (np.float32(12.5), np.float32(np.inf), np.float32(np.nan)):
        for early in (False, True):
            mask = random_mask(rng, n_rows, t)
            a = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, "PLUS", add, "NQZERO", init, early)
            b = reference.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, "PLUS", add, "NQZERO", init, early)
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg=f"mxv init={init} early={early}")
    vi = np.sort(rng.permutation(n_cols)[:30]).astype(np.uint32)  # vxm: v indexes the ROWS of the (n_cols x n_rows) transposed problem below
    AT = random_csr(rng, n_cols, n_rows, 7, t, skew=True)
    ATx = (rng.integers(-40, 400, len(AT[1])) / 8.0).astype(np.float32)
    vx = v[vi].copy()
    vx[0] = np.nan
    a = oracle.vxm_masked(t, n_cols, n_rows, AT[0], AT[1], ATx, np.ones(n_rows, np.float32), vi, vx, "PLUS", add, "ALWAYS")
    b = reference.vxm_masked(t, n_cols, n_rows, AT[0], AT[1], ATx, np.ones(n_rows, np.float32), vi, vx, "PLUS", add, "ALWAYS")
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.isnan(a[1]).any() and not np.isnan(a[1]).all()
    A = random_csr(rng, 23, 31, 5, t)
    B = random_csr(rng, 31, 19, 5, t)
    bx = B[2].astype(np.float32)
    bx[rng.random(len(bx)) < 0.2] = np.nan
    B = (B[0], B[1], bx)
    for init in (np.float32(100.0), np.float32(np.nan)):
        a = oracle.mxm(t, 23, 19, A, B, "PLUS", add, init)
        b = reference.mxm(t, 23, 19, A, B, "PLUS", add, init)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32), err_msg=f"mxm init={init}")
    x = (rng.integers(-400, 400, 77) / 8.0).astype(np.float32)
    x[rng.random(77) < 0.3] = np.nan
    x[0] = np.nan
    for s in (np.float32(0.5), np.float32(np.nan)):
        assert np.array_equal(np.array([oracle.reduce(t, x, add, s)], np.float32).view(np.uint32), np.array([reference.reduce(t, x, add, s)], np.float32).view(np.uint32))
