"""Parity of the HIP masked SpMV (through the kernel-level C ABI) against the oracle on seeded inputs.
Bit-exact for int32 / uint32 and order-free monoids; <= 1e-5 relative for fp32 PLUS / MULT folds (north star)."""
import numpy as np
import pytest

from helpers import (NP, SELECTS, SEMIRINGS_COMMON, SEMIRINGS_INT, SEMIRINGS_ORDERED, assert_same, random_csr,
                     random_mask, random_values)

pytestmark = pytest.mark.gpu

FP32_RTOL = 1e-5


def _init_for(add):
    return {"MIN": 100, "MAX": 0, "MULT": 1, "LAND": 1}.get(add, 0)


def test_reference_unit_case_naive(oracle):
    from gpu_utils import hip_mxv
    # reference tests/test_mxv.cpp:33-89, r pre-loaded with junk to prove every row is overwritten
    Ap = np.array([0, 2, 4, 5, 6], np.uint32)
    Aj = np.array([1, 4, 0, 4, 2, 4], np.uint32)
    Ax = np.array([2, -9, 2, -8, 3, -1], np.int32)
    mask = np.array([1, 0, 1, 0], np.int32)
    v = np.array([3, 0, 3, 0, -1], np.int32)
    r = hip_mxv("INT", 4, 5, Ap, Aj, Ax, mask, v, "MULT", "PLUS", "EQZERO", 0, junk=77)
    assert r.tolist() == [0, 14, 0, 1]


def test_reference_perf_case_banded(oracle):
    from gpu_utils import hip_mxv
    # reference tests/test_mxv.cpp:91-130 at reduced N (banded, 256 nnz/row, every 10th row selected)
    N, K, S = 20000, 256, 10
    Ap = (np.arange(N + 1, dtype=np.uint64) * K).astype(np.uint32)
    Aj = np.sort((np.arange(N)[:, None] + np.arange(K)[None, :]) % N, axis=1).astype(np.uint32).ravel()
    Ax = np.ones(N * K, np.int32)
    mask = (np.arange(N) % S != 0).astype(np.int32)
    v = np.ones(N, np.int32)
    r = hip_mxv("INT", N, N, Ap, Aj, Ax, mask, v, "MULT", "PLUS", "EQZERO", 0)
    assert np.array_equal(r, np.where(np.arange(N) % S, 0, K))


def _semirings(t):
    s = SEMIRINGS_COMMON + SEMIRINGS_ORDERED
    return s + SEMIRINGS_INT if t != "FLOAT" else s


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_all_semiring_cells(oracle, t):
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(101)
    n_rows, n_cols = 3000, 2500
    # skewed degrees, empty rows, rows longer than the lane-per-row limit and two rows longer than the LDS stage
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 9, t, skew=True, long_rows=2, long_len=2400)
    for k, (mul, add) in enumerate(_semirings(t)):
        for early in (False, True):
            sel = SELECTS[(k + early) % len(SELECTS)]
            mask = random_mask(rng, n_rows, t)
            v = random_values(rng, n_cols, t, zeros=0.3)
            init = _init_for(add)
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, early)
            got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, early)
            exact = not (t == "FLOAT" and add in ("PLUS", "MULT") and not early)
            if exact:
                np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=f"{t} {mul}/{add} {sel} early={early}")
            else:
                np.testing.assert_allclose(got, want, rtol=FP32_RTOL, atol=0, err_msg=f"{t} {mul}/{add} {sel}")


@pytest.mark.parametrize("t", ["INT", "FLOAT"])
def test_long_rows_and_row_caps(oracle, t):
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(5)
    # one 20000-long row (peeled, strided from HBM), >1024 consecutive empty rows (row cap), medium rows 33..2000
    n_cols = 30000
    degs = np.concatenate([np.zeros(2500, np.int64), [20000], rng.integers(33, 2000, 40), np.zeros(1500, np.int64),
                           [4096, 4097, 4095, 1, 0, 2047, 2048, 2049], rng.integers(0, 5, 3000)])
    n_rows = len(degs)
    Ap = np.zeros(n_rows + 1, np.uint32)
    Ap[1:] = np.cumsum(degs)
    Aj = np.concatenate([np.sort(rng.permutation(n_cols)[:d]) for d in degs]).astype(np.uint32)
    Ax = random_values(rng, len(Aj), t)
    v = random_values(rng, n_cols, t, zeros=0.1)
    for mul, add, sel in [("MULT", "PLUS", "ALWAYS"), ("PLUS", "MIN", "NQZERO"), ("LAND", "LOR", "EQZERO"), ("MIN", "MAX", "GTZERO")]:
        mask = random_mask(rng, n_rows, t)
        init = _init_for(add)
        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
        got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
        if t == "FLOAT" and add == "PLUS":
            np.testing.assert_allclose(got, want, rtol=FP32_RTOL)
        else:
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=f"{mul}/{add}")


def test_edge_shapes(oracle):
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(9)
    t = "INT"
    # no entries at all: every selected row is init, unselected too
    Ap = np.zeros(6, np.uint32)
    r = hip_mxv(t, 5, 4, Ap, np.zeros(0, np.uint32), np.zeros(0, np.int32), np.ones(5, np.int32), np.ones(4, np.int32), "MULT", "PLUS", "ALWAYS", 7, junk=-1)
    assert r.tolist() == [7] * 5
    # single row, single entry
    r = hip_mxv(t, 1, 1, np.array([0, 1], np.uint32), np.array([0], np.uint32), np.array([3], np.int32), np.array([0], np.int32), np.array([5], np.int32), "MULT", "PLUS", "EQZERO", 1)
    assert r.tolist() == [16]
    # NEVER selects nothing: r == init everywhere, matrix untouched
    Ap, Aj, Ax = random_csr(rng, 500, 300, 8, t)
    r = hip_mxv(t, 500, 300, Ap, Aj, Ax, np.ones(500, np.int32), np.ones(300, np.int32), "MULT", "PLUS", "NEVER", -3)
    assert np.all(r == -3)
    # ragged: a 1-row matrix with 10000 entries and a 10000-row matrix with one entry each
    Aj = np.arange(10000, dtype=np.uint32)
    v = rng.integers(-3, 4, 10000).astype(np.int32)
    Ax = rng.integers(-3, 4, 10000).astype(np.int32)
    r = hip_mxv(t, 1, 10000, np.array([0, 10000], np.uint32), Aj, Ax, np.zeros(1, np.int32), v, "MULT", "PLUS", "EQZERO", 0)
    assert r[0] == int(np.sum(Ax.astype(np.int64) * v))
    r = hip_mxv(t, 10000, 10000, np.arange(10001, dtype=np.uint32), Aj, Ax, np.zeros(10000, np.int32), v, "MULT", "PLUS", "EQZERO", 0)
    assert np.array_equal(r, Ax * v)


def test_unaligned_device_arrays(oracle):
    """Aj / Ax that are not 16-byte aligned (a slice of a larger allocation) take the scalar-load path."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(3)
    t = "FLOAT"
    n_rows, n_cols = 2000, 1500
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 12, t, skew=True)
    mask = random_mask(rng, n_rows, t)
    v = random_values(rng, n_cols, t)
    want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, "MULT", "PLUS", "NQZERO", 0.0)
    dAj = dev(np.concatenate([[0], Aj]).astype(np.uint32))[1:]
    dAx = dev(np.concatenate([[0], Ax]).astype(np.float32))[1:]
    assert dAj.data_ptr() % 16 != 0
    dAp, dm, dv = dev(Ap), dev(mask), dev(v)
    plan = kernels.CsrPlan(n_rows, len(Aj), dAp)
    r = torch.empty(n_rows, dtype=torch.float32, device="cuda")
    kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dm, dv, r, "MULT", "PLUS", "NQZERO", 0.0)
    np.testing.assert_allclose(host(r, np.float32), want, rtol=FP32_RTOL)


def test_sharded_rows_equal_whole(oracle):
    """1-D row-block decomposition: running each row block separately (global column ids) gives the same rows."""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(17)
    t = "INT"
    n = 4000
    Ap, Aj, Ax = random_csr(rng, n, n, 10, t, skew=True)
    mask = random_mask(rng, n, t)
    v = random_values(rng, n, t)
    whole = hip_mxv(t, n, n, Ap, Aj, Ax, mask, v, "PLUS", "MIN", "ALWAYS", 1 << 30)
    parts = []
    for lo, hi in [(0, 37), (37, 1800), (1800, 4000)]:
        s, e = int(Ap[lo]), int(Ap[hi])
        parts.append(hip_mxv(t, hi - lo, n, (Ap[lo:hi + 1] - Ap[lo]).astype(np.uint32), Aj[s:e], Ax[s:e], mask[lo:hi], v, "PLUS", "MIN", "ALWAYS", 1 << 30))
    assert np.array_equal(np.concatenate(parts), whole)
    assert np.array_equal(whole, oracle.mxv_masked(t, n, n, Ap, Aj, Ax, mask, v, "PLUS", "MIN", "ALWAYS", 1 << 30))


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_more_columns_than_lds_slots(oracle, t):
    """n_cols beyond the 40,000 LDS-resident columns of the sliced-ELL copy: its lanes then mix LDS reads (hot columns) and global
    gathers (cold columns) inside one wave instruction; hip_mxv cross-checks that form against the CSR kernels, both against the oracle.
    Skewed column popularity so that a clear hot set exists; rows of 1..700 entries (chunks of 256 + partials) and empty rows."""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(91)
    n_rows, n_cols = 6000, 150000
    degs = rng.integers(0, 40, n_rows)
    degs[rng.integers(0, n_rows, 30)] = rng.integers(257, 700, 30)
    degs[::7] = 0
    Ap = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(degs, out=Ap[1:])
    pop = (rng.pareto(1.2, n_cols) + 1e-3)
    pop /= pop.sum()
    Aj = np.concatenate([np.sort(rng.choice(n_cols, size=d, replace=False, p=pop)) for d in degs if d]).astype(np.uint32)
    Ax = random_values(rng, len(Aj), t)
    v = random_values(rng, n_cols, t, zeros=0.2)
    for mul, add, sel, init in [("MULT", "PLUS", "ALWAYS", 0), ("PLUS", "MIN", "NQZERO", 1 << 20), ("FIRST", "MAX", "EQZERO", 0)]:
        mask = random_mask(rng, n_rows, t)
        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
        got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
        assert_same(got, want, t, what=f"{t} {mul}/{add}/{sel}")


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT"])
def test_sparse_input_vector_form(oracle, t):
    """The BITS form of the sliced-ELL kernel (one LDS bit per column instead of a value; semirings where 0 annihilates op_mult and is
    op_add's identity): constant / varying / empty / dense input vectors, non-zero init, chunked long rows whose chunks may hold no
    product, every select op class.  hip_mxv runs the CSR kernels, the auto-choosing sliced-ELL call and the forced BITS form."""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(101)
    n_rows, n_cols = 5000, 90000
    degs = rng.integers(0, 30, n_rows)
    degs[rng.integers(0, n_rows, 25)] = rng.integers(257, 1500, 25)
    degs[::9] = 0
    Ap = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(degs, out=Ap[1:])
    pop = rng.pareto(1.1, n_cols) + 1e-3
    pop /= pop.sum()
    Aj = np.concatenate([np.sort(rng.choice(n_cols, size=d, replace=False, p=pop)) for d in degs if d]).astype(np.uint32)
    Ax = random_values(rng, len(Aj), t)
    Ax[Ax == 0] = 1
    vectors = {
        "frontier (3 % ones)": (rng.random(n_cols) < 0.03).astype(NP[t]),
        "sparse, varying": np.where(rng.random(n_cols) < 0.05, random_values(rng, n_cols, t), 0).astype(NP[t]),
        "all zero": np.zeros(n_cols, NP[t]),
        "dense constant": np.full(n_cols, 3, NP[t]),
        "dense varying": random_values(rng, n_cols, t, zeros=0.1),
        "one entry": np.eye(1, n_cols, 77, dtype=NP[t])[0] * 5,
    }
    cells = [("BAND", "BOR", "EQZERO", 0), ("MULT", "PLUS", "ALWAYS", 5), ("LAND", "BXOR", "NQZERO", 9), ("MULT", "BOR", "GTZERO", 0)]
    if t == "UINT":
        cells.append(("BAND", "MAX", "ALWAYS", 2))
    for name, v in vectors.items():
        for mul, add, sel, init in cells:
            mask = random_mask(rng, n_rows, t)
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
            got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, junk=123)
            assert_same(got, want, t, what=f"{t} {name} {mul}/{add}/{sel}")


@pytest.mark.gpu
def test_sparse_input_vector_cold_ranks(oracle):
    """more referenced columns than the 1.28 M LDS-resident bits: the coldest ranks read the global bitmap"""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(102)
    n_rows, n_cols = 30000, 1_600_000
    per = 50
    cols = rng.permutation(n_cols)[:n_rows * per].reshape(n_rows, per)  # 1.5 M distinct columns, each referenced once
    cols.sort(axis=1)
    hub = rng.integers(0, 2000, (n_rows, 6))                            # plus a hot set so that the ranking is not degenerate
    rows = [np.unique(np.concatenate([cols[i], hub[i]])) for i in range(n_rows)]
    Ap = np.zeros(n_rows + 1, np.uint32)
    np.cumsum([len(x) for x in rows], out=Ap[1:])
    Aj = np.concatenate(rows).astype(np.uint32)
    Ax = np.ones(len(Aj), np.int32)
    v = (rng.random(n_cols) < 0.2).astype(np.int32)
    mask = random_mask(rng, n_rows, "INT")
    for mul, add, sel, init in [("BAND", "BOR", "EQZERO", 0), ("MULT", "PLUS", "ALWAYS", 1)]:
        want = oracle.mxv_masked("INT", n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
        got = hip_mxv("INT", n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
        assert_same(got, want, "INT", what=f"cold ranks {mul}/{add}")
    # the one-way folds of the cold list (a plain load decides whether the product can still move r[row]): min-plus with weights and
    # varying finite values (int32, fp32) and uint MAX, most entries of a row cold
    for t, mul, add, init in (("INT", "PLUS", "MIN", 1 << 30), ("FLOAT", "PLUS", "MIN", 3.0e38), ("UINT", "MULT", "MAX", 0)):
        w = rng.integers(1, 200, len(Aj)).astype(NP[t])
        vv = np.full(n_cols, 0 if add == "MAX" else init, NP[t])
        k = rng.random(n_cols) < 0.15
        vv[k] = rng.integers(1, 100000, int(k.sum())).astype(NP[t])
        m = random_mask(rng, n_rows, t)
        for sel in ("ALWAYS", "EQZERO"):
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, w, m, vv, mul, add, sel, init)
            got = hip_mxv(t, n_rows, n_cols, Ap, Aj, w, m, vv, mul, add, sel, init, junk=5)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (t, mul, add, sel)


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "FLOAT"])
def test_row_lengths_at_the_slice_boundaries(oracle, t):
    """rows of exactly 255 / 256 / 257 entries (last single-chunk row, first row-exclusive slice), around one 64 x 256 = 16,384-entry
    block of a row-exclusive slice (partly filled last lanes, a second slice of one entry) and a 40,000-entry row; value form, bit form
    and the CSR kernels against the oracle, masks that select / skip exactly those rows"""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(303)
    lens = [255, 256, 257, 319, 320, 321, 511, 512, 513, 16383, 16384, 16385, 16447, 16448, 16449, 40000, 1, 0, 64, 63, 65]
    n_cols = 41000
    n_rows = len(lens) * 3
    degs = np.zeros(n_rows, np.int64)
    degs[::3] = lens                       # the special rows, separated by short / empty ones
    degs[1::3] = rng.integers(0, 6, len(lens))
    Ap = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(degs, out=Ap[1:])
    Aj = np.concatenate([np.sort(rng.choice(n_cols, size=int(d), replace=False)) for d in degs if d]).astype(np.uint32)
    Ax = random_values(rng, len(Aj), t)
    if t == "INT":
        Ax[Ax == 0] = 1
    v_dense = random_values(rng, n_cols, t, zeros=0.1)
    v_sparse = np.where(rng.random(n_cols) < 0.02, 1, 0).astype(NP[t])
    cells = [("MULT", "PLUS", "ALWAYS", 0, v_dense), ("PLUS", "MIN", "NQZERO", 1 << 20, v_dense), ("MULT", "PLUS", "EQZERO", 7, v_sparse)]
    if t == "INT":
        cells += [("BAND", "BOR", "EQZERO", 0, v_sparse), ("BAND", "BOR", "ALWAYS", 3, v_dense)]
    for mul, add, sel, init, v in cells:
        for mask in (random_mask(rng, n_rows, t), np.zeros(n_rows, NP[t]), np.ones(n_rows, NP[t])):
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
            got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, junk=-5)
            assert_same(got, want, t, what=f"{t} {mul}/{add}/{sel}")


@pytest.mark.gpu
def test_fp32_product_monoid_keeps_the_left_fold(oracle):
    """op_add = MULT in fp32: a tree of 600 factors around 2 overflows to inf (and 0 x inf = NaN) where the reference's left fold from
    init 0 stays 0 — no tolerance covers that, so the kernels must fold such rows strictly left to right (found by tools/exp/fuzz_mxv.py)"""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(404)
    n_rows, n_cols = 300, 5000
    degs = rng.integers(1, 40, n_rows)
    degs[::10] = rng.integers(300, 900, len(degs[::10]))
    Ap = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(degs, out=Ap[1:])
    Aj = np.concatenate([np.sort(rng.choice(n_cols, size=int(d), replace=False)) for d in degs]).astype(np.uint32)
    Ax = (rng.integers(24, 64, len(Aj)) / 16.0).astype(np.float32)       # 1.5 .. 4
    v = (rng.integers(16, 48, n_cols) / 16.0).astype(np.float32)          # 1 .. 3
    for init in (0.0, 1.0, 1e-30):
        mask = random_mask(rng, n_rows, "FLOAT")
        want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask, v, "MULT", "MULT", "ALWAYS", init)
        got = hip_mxv("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask, v, "MULT", "MULT", "ALWAYS", init)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), init


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_min_plus_bit_form(oracle, t):
    """min-plus with the bit form of the sliced-ELL kernel: an entry whose v value equals init (the "infinity" of a shortest-path step)
    cannot lower a minimum when no matrix value is negative.  Sparse / dense / constant finite parts of v, long rows, masks; and the
    cases the form must refuse (a negative weight, an init that would wrap) still equal the oracle through the value form."""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(505)
    n_rows, n_cols = 4000, 70000
    degs = rng.integers(0, 25, n_rows)
    degs[rng.integers(0, n_rows, 20)] = rng.integers(257, 3000, 20)
    Ap = np.zeros(n_rows + 1, np.uint32)
    np.cumsum(degs, out=Ap[1:])
    pop = rng.pareto(1.1, n_cols) + 1e-3
    pop /= pop.sum()
    Aj = np.concatenate([np.sort(rng.choice(n_cols, size=int(d), replace=False, p=pop)) for d in degs if d]).astype(np.uint32)
    INF = {"INT": 1 << 30, "UINT": 1 << 31, "FLOAT": np.finfo(np.float32).max}[t]
    w = rng.integers(1, 256, len(Aj)).astype(NP[t])
    for name, frac in (("sparse", 0.02), ("eighth", 0.125), ("dense", 0.9), ("none", 0.0)):
        v = np.full(n_cols, INF, NP[t])
        k = rng.random(n_cols) < frac
        v[k] = rng.integers(0, 5000, int(k.sum())).astype(NP[t])
        for sel in ("ALWAYS", "EQZERO"):
            mask = random_mask(rng, n_rows, t)
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, w, mask, v, "PLUS", "MIN", sel, INF)
            got = hip_mxv(t, n_rows, n_cols, Ap, Aj, w, mask, v, "PLUS", "MIN", sel, INF, junk=7)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (t, name, sel)
    # all finite values equal: the constant case of the form
    v = np.full(n_cols, INF, NP[t])
    v[rng.random(n_cols) < 0.05] = 17
    want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, w, np.zeros(n_rows, NP[t]), v, "PLUS", "MIN", "ALWAYS", INF)
    got = hip_mxv(t, n_rows, n_cols, Ap, Aj, w, np.zeros(n_rows, NP[t]), v, "PLUS", "MIN", "ALWAYS", INF)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # refused: a negative weight (INT / FLOAT), or an init so large that init + weight wraps (UINT)
    if t == "UINT":
        big = np.uint32(0xFFFFFF00)
        v2 = np.where(rng.random(n_cols) < 0.1, 5, big).astype(np.uint32)
        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, w, np.zeros(n_rows, NP[t]), v2, "PLUS", "MIN", "ALWAYS", int(big))
        got = hip_mxv(t, n_rows, n_cols, Ap, Aj, w, np.zeros(n_rows, NP[t]), v2, "PLUS", "MIN", "ALWAYS", int(big))
    else:
        w2 = w.copy()
        w2[::97] = -3
        v2 = np.where(rng.random(n_cols) < 0.1, 5, INF).astype(NP[t])
        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, w2, np.zeros(n_rows, NP[t]), v2, "PLUS", "MIN", "ALWAYS", INF)
        got = hip_mxv(t, n_rows, n_cols, Ap, Aj, w2, np.zeros(n_rows, NP[t]), v2, "PLUS", "MIN", "ALWAYS", INF)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_uniform_matrix_values_structure_only(oracle, t):
    """A matrix whose stored values are all one word (a graph's adjacency matrix) is served by the sliced-ELL bit form without its value
    stream.  Values other than 1, every bit-form semiring family, long rows, both input-vector densities; one odd value switches the
    shortcut off; SPLAHIP_SELL_UNIFORM=0 streams the words again — all equal the oracle."""
    import os
    import torch
    from gpu_utils import dev, hip_mxv
    from spla_amd import kernels
    rng = np.random.default_rng(707)
    n_rows, n_cols = 5000, 90000
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 10, t, skew=True, long_rows=3, long_len=1500)
    fams = [("PLUS", "MIN", 1 << 20)] + ([("BAND", "BOR", 0), ("MULT", "PLUS", 2), ("LAND", "BXOR", 0)] if t != "FLOAT" else [])
    for xval in (1, 6):
        Ax = np.full(len(Aj), xval, NP[t])
        plan = kernels.CsrPlan(n_rows, len(Aj), dev(Ap), n_cols, dev(Aj), dev(Ax))
        assert plan.sell["uniform_values"] is True
        plan.close()
        for mul, add, init in fams:
            for frac in (0.01, 0.6):
                v = np.full(n_cols, init if add == "MIN" else 0, NP[t])
                k = rng.random(n_cols) < frac
                v[k] = rng.integers(1, 9, int(k.sum())).astype(NP[t])
                for sel in ("ALWAYS", "EQZERO"):
                    mask = random_mask(rng, n_rows, t)
                    want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
                    for uni in ("1", "0"):
                        os.environ["SPLAHIP_SELL_UNIFORM"] = uni
                        try:
                            got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, junk=3)
                        finally:
                            os.environ.pop("SPLAHIP_SELL_UNIFORM")
                        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (t, xval, mul, add, frac, sel, uni)
    Ax = np.full(len(Aj), 1, NP[t])
    Ax[len(Aj) // 2] = 2
    plan = kernels.CsrPlan(n_rows, len(Aj), dev(Ap), n_cols, dev(Aj), dev(Ax))
    assert plan.sell["uniform_values"] is False
    plan.close()
    v = (rng.random(n_cols) < 0.05).astype(NP[t])
    mul, add, init = ("PLUS", "MIN", 1 << 20) if t == "FLOAT" else ("MULT", "PLUS", 0)
    if add == "MIN":
        v = np.where(v > 0, v, init).astype(NP[t])
    want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, np.zeros(n_rows, NP[t]), v, mul, add, "ALWAYS", init)
    got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, np.zeros(n_rows, NP[t]), v, mul, add, "ALWAYS", init)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT"])
def test_frontier_form_counts_hits(oracle, t):
    """The frontier form of the sliced-ELL bit form (mxv_sell.hpp: sell_frontier_body): matrix values all one word AND every non-skipped
    value of v one word, so a row's result is fold_repeated(init, product, hits).  Row lengths on both sides of every boundary of its walk
    (quads of 4, the 16-position first group, the 32-load window, 256-entry chunks, row-exclusive slices of 16 Ki entries), hit counts 0 /
    1 / many, the counting (PLUS), parity (BXOR) and idempotent (BOR, MAX, LOR, min-plus) families, non-zero init, ALWAYS and a mask, and
    more columns than the LDS bitmap holds (the cold list folds the rest)."""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(4711)
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 255, 256, 257, 300, 511, 1000, 4095, 16384, 16385, 40000]
    lens = lens + [int(x) for x in rng.integers(0, 40, 400)]
    for n_cols in (60000, 1500000):
        n_rows = len(lens)
        Ap = np.zeros(n_rows + 1, np.uint32)
        cols = []
        for i, ln in enumerate(lens):
            ln = min(ln, n_cols)
            c = np.sort(rng.choice(n_cols, ln, replace=False)) if ln else np.zeros(0, np.int64)
            cols.append(c.astype(np.uint32))
            Ap[i + 1] = Ap[i] + ln
        Aj = np.concatenate(cols)
        for xval, cval in ((1, 1), (3, 5)):
            Ax = np.full(len(Aj), xval, NP[t])
            fams = [("BAND", "BOR", 0), ("MULT", "PLUS", 0), ("MULT", "PLUS", 7), ("LAND", "BXOR", 0), ("LAND", "LOR", 0), ("PLUS", "MIN", 1 << 20)]
            if t == "UINT":
                fams.append(("MULT", "MAX", 0))
            for mul, add, init in fams:
                for frac in (0.0, 0.002, 0.5):
                    skip = init if add == "MIN" else 0
                    v = np.full(n_cols, skip, NP[t])
                    v[rng.random(n_cols) < frac] = cval
                    for sel in ("ALWAYS", "NQZERO"):
                        mask = random_mask(rng, n_rows, t)
                        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
                        got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, junk=9)
                        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (t, n_cols, xval, cval, mul, add, init, frac, sel)


@pytest.mark.gpu
@pytest.mark.parametrize("add", ["MIN", "MAX"])
def test_fp32_min_max_keyed_form(oracle, add):
    """fp32 MIN / MAX over op_mult = PLUS on the column-ordered blocks (the reference's own SSSP semiring, src/algorithm.cpp:208-213): the LDS
    slice holds order-preserving integer keys.  Infinities of both signs, NaN in v (a NaN product is never picked by std::min / std::max as
    the second argument) through ALL FIVE kernel forms, negative weights, a NaN / infinite init, and a matrix holding -0.0f (the keyed form
    must decline) — all bit for bit."""
    import os
    import torch
    from gpu_utils import dev, hip_mxv, host
    from spla_amd import kernels
    rng = np.random.default_rng(808 + (add == "MAX"))
    n_rows, n_cols = 3500, 4000
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 11, "FLOAT", skew=True, long_rows=2, long_len=1500)
    nnz = len(Aj)
    w = (rng.integers(-40, 400, nnz) / 8.0).astype(np.float32)
    v = (rng.integers(-500, 5000, n_cols) / 4.0).astype(np.float32)
    v[rng.random(n_cols) < 0.3] = np.inf
    v[rng.random(n_cols) < 0.02] = -np.inf
    v[rng.random(n_cols) < 0.02] = -0.0
    v_nan = v.copy()
    v_nan[rng.random(n_cols) < 0.03] = np.nan
    dAp, dAj, dw = dev(Ap), dev(Aj), dev(w)
    plans = []
    for small in (False, True):
        if small:
            os.environ["SPLAHIP_COB_HUB_MIN_DEGREE"] = "3"
        p = kernels.CsrPlan(n_rows, nnz, dAp)
        assert p.attach_cob(n_cols, dAp, dAj, dw, row_bits=8 if small else 0, tile_chunks=2 if small else 0)
        os.environ.pop("SPLAHIP_COB_HUB_MIN_DEGREE", None)
        assert "integer keys" in kernels.mxv_variant("FLOAT", "PLUS", add, False, p)
        plans.append(p)
    for init in (np.float32(np.inf), np.float32(-np.inf), np.float32(12.5), np.float32(np.nan), np.float32(-0.0)):
        for sel in ("ALWAYS", "NQZERO"):
            mask = random_mask(rng, n_rows, "FLOAT")
            want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, w, mask, v, "PLUS", add, sel, init)
            got = hip_mxv("FLOAT", n_rows, n_cols, Ap, Aj, w, mask, v, "PLUS", add, sel, init, junk=3.0)  # every layout, cross-checked
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (add, init, sel)
            # NaN among the products: the reference's sequential fold never picks a NaN second argument (src/op.cpp:151-155) and a NaN init
            # sticks.  Every form — CSR wave tiles (in-order tree), sliced-ELL (lane-strided partials), the blocks' keyed form — must agree
            # with it bit for bit (Reassoc / canonical_add, ops.hpp)
            want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, w, mask, v_nan, "PLUS", add, sel, init)
            got = hip_mxv("FLOAT", n_rows, n_cols, Ap, Aj, w, mask, v_nan, "PLUS", add, sel, init, junk=3.0, exact=True)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (add, init, sel, "NaN products, all five forms")
            for p in plans:
                r = torch.full((n_rows,), 3.0, dtype=torch.float32, device="cuda")
                kernels.mxv_masked(p, "FLOAT", n_cols, dAp, dAj, dw, dev(mask), dev(v_nan), r, "PLUS", add, sel, init)
                torch.cuda.synchronize()
                assert np.array_equal(host(r, np.float32).view(np.uint32), want.view(np.uint32)), (add, init, sel, "NaN products")
    for p in plans:
        p.close()
    # a -0.0f among the matrix values: key order would put -0 below +0 where the reference keeps the first zero — the form declines
    w2 = w.copy()
    w2[::53] = -0.0
    v2 = np.where(rng.random(n_cols) < 0.5, np.float32(-0.0), np.float32(0.0)).astype(np.float32)
    want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, w2, np.zeros(n_rows, np.float32), v2, "PLUS", add, "ALWAYS", np.float32(5.0))
    got = hip_mxv("FLOAT", n_rows, n_cols, Ap, Aj, w2, np.zeros(n_rows, np.float32), v2, "PLUS", add, "ALWAYS", np.float32(5.0))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_cob_plan_reused_across_semirings(oracle):
    """One plan, many calls: the shared accumulators of the ranges cut into several pieces rest at the identity of the op_add that used them
    last and the tickets at zero — alternate semirings (and repeat each) on the same plan and compare every result with the oracle."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(4711)
    n_rows, n_cols = 6000, 5000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 14, "INT", skew=True, long_rows=3, long_len=3000)
    dAp, dAj, dAx = dev(Ap), dev(Aj), dev(Ax)
    import os
    os.environ["SPLAHIP_COB_HUB_MIN_DEGREE"] = "4"
    plan = kernels.CsrPlan(n_rows, len(Aj), dAp)
    info = plan.attach_cob(n_cols, dAp, dAj, dAx, row_bits=8, tile_chunks=3)
    os.environ.pop("SPLAHIP_COB_HUB_MIN_DEGREE")
    assert info and info["ranges_in_pieces"] > 0 and info["hub_rows"] > 0, info
    r = torch.empty(n_rows, dtype=torch.int32, device="cuda")
    calls = [("MULT", "PLUS", 0), ("PLUS", "MIN", 1 << 20), ("PLUS", "MIN", 1 << 20), ("BAND", "BOR", 0), ("MULT", "PLUS", 7), ("MULT", "MAX", -5),
             ("BOR", "BAND", -1), ("MULT", "BXOR", 0), ("LAND", "LOR", 0), ("LOR", "LAND", 1), ("MULT", "PLUS", 0)]
    for it, (mul, add, init) in enumerate(calls * 2):
        mask = random_mask(rng, n_rows, "INT")
        v = random_values(rng, n_cols, "INT", zeros=0.2)
        sel = SELECTS[it % len(SELECTS)]
        assert "mxv_cob" in kernels.mxv_variant("INT", mul, add, False, plan)
        r.fill_(-77)
        kernels.mxv_masked(plan, "INT", n_cols, dAp, dAj, dAx, dev(mask), dev(v), r, mul, add, sel, init)
        torch.cuda.synchronize()
        want = oracle.mxv_masked("INT", n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
        assert np.array_equal(host(r, np.int32), want), (it, mul, add, sel)
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [{"SPLAHIP_COB_HUB_FINISH": "0"}, {"SPLAHIP_COB_HUB_FINISH": "2", "SPLAHIP_COB_FIN_SLOTS": "256"}, {"SPLAHIP_COB_ORDER": "0"},
                                   {"SPLAHIP_COB_ORDER": "1"}, {"SPLAHIP_COB_PER_CU": "2"}, {"SPLAHIP_COB_TAPER": "1", "SPLAHIP_COB_LEVELS": "2"},
                                   {"SPLAHIP_COB_NAT_SLOTS": "1024", "SPLAHIP_COB_HUB_FINISH": "2"}, {"SPLAHIP_COB_HUB_ROW_ORDER": "0", "SPLAHIP_COB_HUB_RANGES": "2"}])
def test_cob_schedule_and_geometry_knobs(oracle, knobs):
    """The selectable forms of the column-ordered blocks that are not the default (cob.hip: who finishes the hub rows — the range's last piece, the second
    launch, finisher workgroups inside the main launch —, launch orders, two workgroups per CU, the equal-cost taper, natural ranges larger than hub ranges,
    degree-ordered hub slots, few hub ranges): same results as the oracle, bit for bit on int32 (PLUS, MIN, BOR) and within 1e-5 on fp32 PLUS, twice per plan
    (accumulators, tickets and finisher counters must be back at rest after a call)."""
    import os

    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(99)
    n_rows, n_cols = 9000, 7000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 12, "INT", skew=True, long_rows=3, long_len=2500)
    Axf = rng.random(len(Aj)).astype(np.float32)
    dAp, dAj = dev(Ap), dev(Aj)
    env = dict(knobs, SPLAHIP_COB_HUB_MIN_DEGREE="4")
    for k, val in env.items():
        os.environ[k] = val
    try:
        for t, ax in (("INT", Ax), ("FLOAT", Axf)):
            dAx = dev(ax)
            plan = kernels.CsrPlan(n_rows, len(Aj), dAp)
            info = plan.attach_cob(n_cols, dAp, dAj, dAx, row_bits=8, tile_chunks=3)
            assert info and info["ranges_in_pieces"] > 0 and info["hub_rows"] > 0, info
            r = torch.empty(n_rows, dtype=kernels.torch_dtype(t), device="cuda")
            calls = [("MULT", "PLUS", 0), ("PLUS", "MIN", 1 << 20), ("BAND", "BOR", 0)] if t == "INT" else [("MULT", "PLUS", 0.0), ("PLUS", "MIN", 1e30)]
            for it, (mul, add, init) in enumerate(calls * 2):
                mask = random_mask(rng, n_rows, t)
                v = random_values(rng, n_cols, t, zeros=0.2) if t == "INT" else rng.random(n_cols).astype(np.float32)
                sel = SELECTS[it % len(SELECTS)]
                assert "mxv_cob" in kernels.mxv_variant(t, mul, add, False, plan)
                r.fill_(-77)
                kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dev(mask), dev(v), r, mul, add, sel, init)
                torch.cuda.synchronize()
                want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, ax, mask, v, mul, add, sel, init)
                got = host(r, NP[t])
                if t == "INT" or add != "PLUS":
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (knobs, t, it, mul, add, sel)
                else:
                    assert np.allclose(got, want.view(np.float32), rtol=FP32_RTOL, atol=1e-6), (knobs, t, it, mul, add, sel)
            plan.close()
    finally:
        for k in env:
            os.environ.pop(k)


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "FLOAT"])
def test_masked_call_takes_the_side_its_mask_favours(oracle, t):
    """Every masked call over a plan that holds the column-ordered blocks enqueues a probe of its mask (cob_mask_probe_kernel: how many ENTRIES does the selection
    cover) whose verdict lands in pinned host memory; the host enqueues the side the latest finished probe favours: the selected-rows kernels (mxv_rows.hpp — they
    read selected rows only, as the reference skips unselected rows: src/cpu/cpu_mxv.hpp:89-92) or the blocks, which stream the whole matrix.  Both sides give the
    oracle's result for any mask (long rows included: the hub rows of this matrix hold thousands of entries); the second call with a mask runs the favoured side."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(2024)
    n_rows, n_cols = 9000, 7000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 12, t, skew=True, long_rows=4, long_len=3500)
    if t == "FLOAT":
        Ax = (rng.integers(1, 64, len(Aj)) / 8.0).astype(np.float32)  # (sums of these are exact in any order)
        v = (rng.integers(0, 32, n_cols) / 4.0).astype(np.float32)
    else:
        v = random_values(rng, n_cols, t, zeros=0.2)
    deg = np.diff(Ap.astype(np.int64))
    dAp, dAj, dAx, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(v, NP[t]))
    import os
    os.environ["SPLAHIP_COB_HUB_MIN_DEGREE"] = "20"
    plan = kernels.CsrPlan(n_rows, len(Aj), dAp)
    info = plan.attach_cob(n_cols, dAp, dAj, dAx, row_bits=8, tile_chunks=3)
    os.environ.pop("SPLAHIP_COB_HUB_MIN_DEGREE")
    assert info and info["hub_ranges"] >= 2
    hubs = deg >= 20
    masks = {
        "5 % of the rows": (rng.random(n_rows) < 0.05),
        "60 % of the rows": (rng.random(n_rows) < 0.6),
        "no hub row": ~hubs & (rng.random(n_rows) < 0.9),
        "only hub rows": hubs,
        "nothing": np.zeros(n_rows, bool),
    }
    seen = set()
    for name, selrows in masks.items():
        mask = np.where(selrows, 3, 0).astype(NP[t])
        for mul, add, init in (("MULT", "PLUS", 0), ("PLUS", "MIN", 1000)):
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, "NQZERO", init)
            for attempt in (0, 1):  # the first call with a new mask runs the side the PREVIOUS mask's probe favoured, the second the side this mask's does
                r = torch.full((n_rows,), 77, dtype=kernels.torch_dtype(t), device="cuda")
                kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dev(mask), dv, r, mul, add, "NQZERO", init)
                torch.cuda.synchronize()
                probe = plan.mask_probe()
                got = host(r, NP[t])
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (t, name, mul, add, attempt, probe)
            assert probe["selected_entries"] == int(deg[selrows].sum()), (name, probe)  # (a matrix this small is probed row by row; large ones sample 2048-row groups)
            seen.add(probe["served_by_rows"])
            if name in ("5 % of the rows", "nothing"):
                assert probe["served_by_rows"], (name, probe)
            if name in ("60 % of the rows", "only hub rows"):
                assert not probe["served_by_rows"], (name, probe)
    assert seen == {True, False}
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "FLOAT"])
def test_hot_column_relabelling(oracle, t, monkeypatch):
    """A matrix whose column labels carry no locality gets its hot columns relabelled inside the column-ordered blocks (cob.hip: the columns referenced most take
    the ids 0 .. K-1, a call gathers from [v[hot columns] | v]).  Forced on here (SPLAHIP_COB_RELABEL=2, every column referenced >= 8 times is hot) on a matrix
    of 70 000 randomly labelled columns: mxv in the value form and the sparse-input-vector forms, masked and unmasked, and the pull form of vxm over the copy of
    the transpose — all against the oracle."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(77)
    n_rows, n_cols = 6000, 70000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 40, t, skew=True, long_rows=3, long_len=5000)
    perm = rng.permutation(n_cols).astype(np.uint32)  # scatter the skewed columns over the id space
    Aj = perm[Aj]
    for i in range(n_rows):  # rows stay column-sorted (the reference's CSR invariant)
        a, b = int(Ap[i]), int(Ap[i + 1])
        o = np.argsort(Aj[a:b], kind="stable")
        Aj[a:b], Ax[a:b] = Aj[a:b][o], Ax[a:b][o]
    if t == "FLOAT":
        Ax = (rng.integers(1, 64, len(Aj)) / 8.0).astype(np.float32)
        v = (rng.integers(0, 32, n_cols) / 4.0).astype(np.float32)
    else:
        v = random_values(rng, n_cols, t, zeros=0.5)
    dAp, dAj, dAx, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(v, NP[t]))
    monkeypatch.setenv("SPLAHIP_COB_RELABEL", "2")
    monkeypatch.setenv("SPLAHIP_COB_RELABEL_MIN", "8")
    monkeypatch.setenv("SPLAHIP_COB_HUB_MIN_DEGREE", "60")
    monkeypatch.setenv("SPLAHIP_MASK_PROBE", "0")
    plan = kernels.CsrPlan(n_rows, len(Aj), dAp)
    info = plan.attach_cob(n_cols, dAp, dAj, dAx, row_bits=9, tile_chunks=4)
    assert info and info["relabelled_hot_columns"] >= 64, info
    cells = [("MULT", "PLUS", 0), ("PLUS", "MIN", 1 << 20), ("BAND", "BOR", 0), ("LAND", "LOR", 0)] if t == "INT" else [("MULT", "PLUS", 0.0), ("PLUS", "MIN", 1e9), ("LAND", "LOR", 0.0)]
    for mul, add, init in cells:
        for sel in ("ALWAYS", "NQZERO"):
            mask = random_mask(rng, n_rows, t)
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
            r = torch.full((n_rows,), 5, dtype=kernels.torch_dtype(t), device="cuda")
            kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dev(mask), dv, r, mul, add, sel, init)
            torch.cuda.synchronize()
            assert "cob" in kernels.mxv_variant(t, mul, add, False, plan)
            if t == "FLOAT" and add == "PLUS":  # (the blocks fold a row's products in column-block order: the stated 1e-5 of the reference's left fold)
                np.testing.assert_allclose(host(r, NP[t]), want, rtol=1e-5, atol=0, err_msg=f"{t} {mul}/{add} {sel}")
            else:
                assert np.array_equal(host(r, NP[t]).view(np.uint32), want.view(np.uint32)), (t, mul, add, sel)
    plan.close()
    # vxm: the pull form over the relabelled copy of the transposed matrix
    Tp, Tj, Tx = kernels.csr_transpose(n_rows, n_cols, dAp, dAj, dAx)
    planT = kernels.CsrPlan(n_cols, len(Aj), Tp)
    # (the transposed matrix has n_rows = 6000 columns: below the relabelling's floor — so transpose a tall copy instead: columns of M^T = rows of M)
    planT.attach_cob(n_rows, Tp, Tj, Tx)
    vi = np.arange(n_rows, dtype=np.uint32)
    vx = (rng.integers(0, 32, n_rows) / 4.0).astype(np.float32) if t == "FLOAT" else random_values(rng, n_rows, t, zeros=0.3)
    maskc = random_mask(rng, n_cols, t)
    want = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, maskc, vi, vx, "MULT", "PLUS", "NQZERO")
    ri, rx = kernels.vxm_pull(planT, t, n_rows, n_cols, Tp, Tj, Tx, dev(maskc), dev(vi), dev(np.asarray(vx, NP[t])), "MULT", "PLUS", "NQZERO", identity=True)
    assert np.array_equal(host(ri, np.uint32), want[0])
    if t == "FLOAT":
        np.testing.assert_allclose(host(rx, NP[t]), want[1], rtol=1e-5, atol=0)
    else:
        assert np.array_equal(host(rx, NP[t]).view(np.uint32), want[1].view(np.uint32))
    planT.close()


def test_deterministic_setting_at_the_kernel_abi(oracle):
    """splahip_set_deterministic (include/spla_hip.h): per host thread; an fp32 PLUS call then leaves the column-ordered blocks for the CSR kernels (the variant
    query says so), twice bit-identical; MIN / MAX (exact in every form) and integer calls keep the blocks; the previous setting is returned."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib, kernels
    rng = np.random.default_rng(4242)
    n_rows, n_cols = 9000, 7000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 10, "FLOAT", skew=True, long_rows=2, long_len=5000)
    mask = random_mask(rng, n_rows, "FLOAT")
    v = random_values(rng, n_cols, "FLOAT", zeros=0.2)
    dAp, dAj, dAx, dm, dv = dev(Ap), dev(Aj), dev(Ax), dev(mask), dev(v)
    plan = kernels.CsrPlan(n_rows, len(Aj), dAp)
    assert plan.attach_cob(n_cols, dAp, dAj, dAx) is not None
    bare = kernels.CsrPlan(n_rows, len(Aj), dAp)
    assert _lib.call("splahip_set_deterministic", 1) == 0
    try:
        assert "mxv_cob" not in kernels.mxv_variant("FLOAT", "MULT", "PLUS", False, plan)
        assert "mxv_cob" in kernels.mxv_variant("FLOAT", "PLUS", "MIN", False, plan)
        assert "mxv_cob" in kernels.mxv_variant("INT", "MULT", "PLUS", False, plan)
        outs = []
        for p in (plan, plan, bare):
            r = torch.full((n_rows,), -3.0, dtype=torch.float32, device="cuda")
            kernels.mxv_masked(p, "FLOAT", n_cols, dAp, dAj, dAx, dm, dv, r, "MULT", "PLUS", "EQZERO", 0.5)
            torch.cuda.synchronize()
            outs.append(host(r, np.float32))
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)) and np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))
        want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask, v, "MULT", "PLUS", "EQZERO", 0.5)
        assert_same(outs[0], want, "FLOAT", what="deterministic fp32 MULT/PLUS")
    finally:
        assert _lib.call("splahip_set_deterministic", 0) == 1
    assert "mxv_cob" in kernels.mxv_variant("FLOAT", "MULT", "PLUS", False, plan)
    plan.close()
    bare.close()


@pytest.mark.parametrize("copy", ["sell", "cob"])
def test_first_hit_form_of_a_bfs_like_step(oracle, copy):
    """A BFS-like step (BAND / MULT / LAND products folded by BOR / uint MAX) over a structure-only matrix with a frontier vector (one non-zero word) is settled
    row by row at the first hit (mxv_first_hit.hpp) — taken from the second call on (the host follows the verdict of the latest FINISHED pass over v), only
    while the sampled walk mostly ends within a row's first entries, and correct when the hint is stale: a vector that stops being uniform is folded completely
    by the same launch.  Rows longer than 64 entries (finished by a whole wave) with and without a hit, non-zero init, an empty vector, masks."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(909)
    n_rows, n_cols = 6000, 5000
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 9, "INT", skew=True, long_rows=6, long_len=2500)
    longest = int(np.argmax(np.diff(Ap.astype(np.int64))))
    for t, mul, add, xval in (("INT", "BAND", "BOR", 1), ("UINT", "MULT", "MAX", 3), ("INT", "LAND", "BOR", 5)):
        Ax = np.full(len(Aj), xval, NP[t])
        dAp, dAj, dAx = dev(Ap), dev(Aj), dev(Ax)
        plan = kernels.CsrPlan(n_rows, len(Aj), dAp, n_cols, dAj, dAx) if copy == "sell" else kernels.CsrPlan(n_rows, len(Aj), dAp)
        if copy == "cob":
            assert plan.attach_cob(n_cols, dAp, dAj, dAx) is not None

        def call(v, mask, sel, init):
            r = torch.full((n_rows,), 91, dtype=kernels.torch_dtype(t), device="cuda")
            kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dev(mask), dev(v), r, mul, add, sel, init, False)
            torch.cuda.synchronize()
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
            assert np.array_equal(host(r, NP[t]), want), (t, mul, add, sel, init, plan.first_hit())
            return plan.first_hit()

        depth = (rng.random(n_rows) < 0.4).astype(NP[t]) * 2
        front = np.where(rng.random(n_cols) < 0.7, 1, 0).astype(NP[t])       # a dense frontier: nearly every row is settled within its first entries
        first = call(front, depth, "EQZERO", 0)
        assert not first["taken"]                                              # no finished pass over a vector of this plan yet
        second = call(front, depth, "EQZERO", 0)
        assert second["taken"] and second["uniform"] and second["nonzeros"] == int(front.sum()) and second["tail_pct"] <= 15, second
        assert "first_hit" in kernels.mxv_variant(t, mul, add, False, plan)
        for sel, init in (("ALWAYS", 0), ("NQZERO", 4), ("EQZERO", 1)):       # (an init that already holds the product's bits: nothing moves)
            assert call(front * 6, depth, sel, init)["taken"]                  # another frontier word
        lone = np.zeros(n_cols, NP[t])
        lone[Aj[Ap[longest] + 2000]] = 1                                       # one column, deep inside the longest row: found by the wave's walk, not by a lane's
        stale = call(lone, np.zeros(n_rows, NP[t]), "EQZERO", 0)               # (hint: the dense frontier of the call before)
        assert stale["taken"] and stale["tail_pct"] > 50, stale
        assert not call(lone, np.zeros(n_rows, NP[t]), "EQZERO", 0)["taken"]   # a frontier that leaves most rows without a hit goes back to the layout copy
        assert not call(front, depth, "EQZERO", 0)["taken"]
        assert call(front, depth, "EQZERO", 0)["taken"]
        mixed = np.where(rng.random(n_cols) < 0.7, rng.integers(1, 8, n_cols), 0).astype(NP[t])
        stale = call(mixed, depth, "EQZERO", 0)                                 # not a frontier: the launch sees it and folds every row completely
        assert stale["taken"] and stale["uniform"] is False
        assert not call(mixed, depth, "EQZERO", 0)["taken"]
        empty = np.zeros(n_cols, NP[t])
        call(empty, depth, "EQZERO", 7)
        call(empty, depth, "EQZERO", 7)
        early = torch.full((n_rows,), 91, dtype=kernels.torch_dtype(t), device="cuda")
        kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dev(depth), dev(front), early, mul, add, "EQZERO", 0, True)   # early_exit = true: same rule, same result
        torch.cuda.synchronize()
        assert np.array_equal(host(early, NP[t]), oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, depth, front, mul, add, "EQZERO", 0, True))
        # an early_exit call always takes this kernel (no hint needed); a vector of several words is folded with the reference's early-exit rule (a row stops behind the
        # product that moves its sum off init: observable with BOR over different words), a sparse frontier and an init that already holds the bits included
        for vec, init in ((mixed, 0), (mixed, 3), (lone, 0), (empty, 0), (front, 1)):
            early.fill_(91)
            kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dev(depth), dev(vec), early, mul, add, "EQZERO", init, True)
            torch.cuda.synchronize()
            assert plan.first_hit()["taken"]
            assert np.array_equal(host(early, NP[t]), oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, depth, vec, mul, add, "EQZERO", init, True)), (t, mul, add, init)
        plan.close()


def test_reference_order_setting_gives_the_reference_fold_bit_for_bit(oracle):
    """splahip_set_reference_order (include/spla_hip.h): an fp32 PLUS fold of mxv and of the pull form of vxm adds every row's products
    strictly left to right from init in entry order — the reference's loop, src/cpu/cpu_mxv.hpp:94-99 / cpu_vxm.hpp:101-113 — so the result is the ORACLE'S
    WORD on every row: rows folded by one lane out of LDS (up to 508 entries), rows of several segment items (one wavefront per row, a chain of dependent
    additions; 509 ... 7000 entries here), unselected rows, empty rows; whatever copies the plan holds (blocks, sliced-ELL: both stand aside); random-magnitude
    values so that every re-association would show.  The setting is per host thread and returns its previous value."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib, kernels
    rng = np.random.default_rng(606)
    n_rows, n_cols = 12000, 9000
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 12, "FLOAT", skew=True, long_rows=6, long_len=7000)
    deg = np.diff(Ap.astype(np.int64))
    assert deg.max() >= 7000 and (deg > 1020).sum() >= 2
    nnz = len(Aj)
    Ax = ((rng.random(nnz) - 0.3) * np.exp(rng.normal(0, 3, nnz))).astype(np.float32)   # mixed signs and magnitudes: a different order is a different word
    v = ((rng.random(n_cols) - 0.4) * np.exp(rng.normal(0, 2, n_cols))).astype(np.float32)
    mask = random_mask(rng, n_rows, "FLOAT")
    dAp, dAj, dAx, dm, dv = dev(Ap), dev(Aj), dev(Ax), dev(mask), dev(v)
    plans = {"csr": kernels.CsrPlan(n_rows, nnz, dAp), "cob": kernels.CsrPlan(n_rows, nnz, dAp), "sell": kernels.CsrPlan(n_rows, nnz, dAp, n_cols, dAj, dAx)}
    assert plans["cob"].attach_cob(n_cols, dAp, dAj, dAx) is not None
    assert _lib.call("splahip_set_reference_order", 1) == 0
    try:
        assert "mxv_ordered" in kernels.mxv_variant("FLOAT", "MULT", "PLUS", False, plans["cob"])
        assert "mxv_cob" in kernels.mxv_variant("FLOAT", "PLUS", "MIN", False, plans["cob"])      # exact in every form: stays
        assert "mxv_cob" in kernels.mxv_variant("INT", "MULT", "PLUS", False, plans["cob"])
        for (mul, sel, init) in [("MULT", "ALWAYS", 0.0), ("MULT", "NQZERO", 0.625), ("PLUS", "EQZERO", -3.5), ("MIN", "GTZERO", 1e-3), ("SECOND", "ALWAYS", 0.0)]:
            want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, "PLUS", sel, init)
            for name, p in plans.items():
                r = torch.full((n_rows,), -7.0, dtype=torch.float32, device="cuda")
                kernels.mxv_masked(p, "FLOAT", n_cols, dAp, dAj, dAx, dm, dv, r, mul, "PLUS", sel, init)
                torch.cuda.synchronize()
                got = host(r, np.float32)
                bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
                assert len(bad) == 0, f"{mul}/PLUS {sel} on the {name} plan: {len(bad)} rows differ from the oracle's word, first row {bad[0]} of {deg[bad[0]]} entries: {got[bad[0]]!r} vs {want[bad[0]]!r}"
    finally:
        assert _lib.call("splahip_set_reference_order", 0) == 1
    # the default mode re-associates: on these values it must round differently somewhere (otherwise this test proves nothing)
    want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask, v, "MULT", "PLUS", "ALWAYS", 0.0)
    r = torch.empty(n_rows, dtype=torch.float32, device="cuda")
    kernels.mxv_masked(plans["cob"], "FLOAT", n_cols, dAp, dAj, dAx, dm, dv, r, "MULT", "PLUS", "ALWAYS", 0.0)
    torch.cuda.synchronize()
    assert (host(r, np.float32).view(np.uint32) != want.view(np.uint32)).any(), "the default fold was expected to round differently somewhere on these values"
    for p in plans.values():
        p.close()


def test_reference_order_setting_vxm_pull_bit_for_bit(oracle):
    """The same setting for the pull form of exec_vxm_masked (the PageRank step of configs[2]): every output index's products are added in the order of v's
    indices from the first product (src/cpu/cpu_vxm.hpp:101-113) — dense v (every index stored) and a sparse v (presence bitmap), a mask, long columns."""
    from gpu_utils import dev, host
    from spla_amd import _lib, kernels
    rng = np.random.default_rng(607)
    n_rows, n_cols = 7000, 8000
    # long COLUMNS of M are long rows of M^T: build M^T with long rows, M = its transpose
    Tp_h, Tj_h, _ = random_csr(rng, n_cols, n_rows, 9, "FLOAT", skew=True, long_rows=5, long_len=4000)
    nnz = len(Tj_h)
    Tx_h = ((rng.random(nnz) - 0.3) * np.exp(rng.normal(0, 3, nnz))).astype(np.float32)
    dTp, dTj, dTx = dev(Tp_h), dev(Tj_h), dev(Tx_h)
    Ap_d, Aj_d, Ax_d = kernels.csr_transpose(n_cols, n_rows, dTp, dTj, dTx)               # M: n_rows x n_cols
    Ap, Aj, Ax = host(Ap_d, np.uint32), host(Aj_d, np.uint32), host(Ax_d, np.float32)
    Tp2, Tj2, Tx2 = kernels.csr_transpose(n_rows, n_cols, Ap_d, Aj_d, Ax_d)               # M^T as the library builds it (stable: entries by ascending row of M)
    mask = random_mask(rng, n_cols, "FLOAT")
    for dense in (True, False):
        if dense:
            vi = np.arange(n_rows, dtype=np.uint32)
        else:
            vi = np.sort(rng.permutation(n_rows)[: n_rows // 3]).astype(np.uint32)
        vx = ((rng.random(len(vi)) - 0.4) * np.exp(rng.normal(0, 2, len(vi)))).astype(np.float32)
        for sel in ("ALWAYS", "NQZERO"):
            wi, wx = oracle.vxm_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, "MULT", "PLUS", sel)
            plan_T = kernels.CsrPlan(n_cols, nnz, Tp2)
            assert plan_T.attach_cob(n_rows, Tp2, Tj2, Tx2) is not None
            assert _lib.call("splahip_set_reference_order", 1) == 0
            try:
                ri, rx = kernels.vxm_pull(plan_T, "FLOAT", n_rows, n_cols, Tp2, Tj2, Tx2, dev(mask), dev(vi), dev(vx), "MULT", "PLUS", sel, identity=dense)
            finally:
                assert _lib.call("splahip_set_reference_order", 0) == 1
            assert np.array_equal(host(ri, np.uint32), wi), f"dense={dense} {sel}: structure"
            got = host(rx, np.float32)
            bad = np.nonzero(got.view(np.uint32) != wx.view(np.uint32))[0]
            assert len(bad) == 0, f"dense={dense} {sel}: {len(bad)} outputs differ from the oracle's word, first {got[bad[0]]!r} vs {wx[bad[0]]!r}"
            plan_T.close()


@pytest.mark.gpu
@pytest.mark.parametrize("t,compact_blocks", [("INT", None), ("UINT", None), ("FLOAT", None), ("FLOAT", "8"), ("INT", "24")])
def test_selected_rows_kernels_forced_on_every_mask_shape(oracle, t, compact_blocks, monkeypatch):
    """The selected-rows path (mxv_rows.hpp, round 6: short rows in 64-row batches with a lane per row, medium rows and the long rows' segment items a wavefront each,
    the long rows listed from the plan) forced on whatever the mask (the probe's weights lowered): random 10 % / 90 % of the rows, every 10th row (the reference's
    own perf test, tests/test_mxv.cpp:91-130), long blocks of consecutive rows, only the long rows, only empty rows, one row, everything — on a matrix of 70 003 rows
    (not a multiple of 8 or of 2048) with rows of every length class: empty, short (<= 32), medium (33 .. 508), rows of one segment item (509 .. 1020) and of
    several.  int / uint bit-identical to the oracle, fp32 sums too (the values are small multiples of 1/8: every order gives the same word)."""
    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    monkeypatch.setenv("SPLAHIP_MASK_ROWS_X10", "1")
    monkeypatch.setenv("SPLAHIP_MASK_LONG_X10", "1")
    if compact_blocks:  # few compaction workgroups: the lists' eight areas of the row space are all in use on this small matrix (by default: the first one only)
        monkeypatch.setenv("SPLAHIP_ROWS_COMPACT_BLOCKS", compact_blocks)
    rng = np.random.default_rng(9091)
    n_rows, n_cols = 70003, 50000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 9, t, skew=True, long_rows=0, long_len=0)
    deg = np.diff(Ap.astype(np.int64))
    # plant rows of the other length classes: rebuild with chosen degrees
    want_deg = deg.copy()
    for d, cnt in ((33, 40), (200, 40), (508, 20), (509, 10), (1020, 10), (1021, 6), (3000, 6), (9000, 3)):
        want_deg[rng.choice(n_rows, cnt, replace=False)] = d
    want_deg[-1] = 700  # (the matrix's last row is a long one; the row before it is empty)
    want_deg[-2] = 0
    Ap = np.zeros(n_rows + 1, np.uint32)
    Ap[1:] = np.cumsum(want_deg)
    nnz = int(Ap[-1])
    Aj = np.empty(nnz, np.uint32)
    for i in np.nonzero(want_deg)[0]:
        Aj[Ap[i]:Ap[i + 1]] = np.sort(rng.choice(n_cols, int(want_deg[i]), replace=False))
    Ax = random_values(rng, nnz, t)
    if t == "FLOAT":
        Ax = (rng.integers(1, 64, nnz) / 8.0).astype(np.float32)
        v = (rng.integers(0, 32, n_cols) / 4.0).astype(np.float32)  # (sums of these are exact in fp32 in any order up to ~2^24 / (8 * 8 * 4) terms: thousands)
    else:
        v = random_values(rng, n_cols, t, zeros=0.2)
    deg = want_deg
    dAp, dAj, dAx, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(v, NP[t]))
    plan = kernels.CsrPlan(n_rows, nnz, dAp)
    assert plan.attach_cob(n_cols, dAp, dAj, dAx) is not None
    blocks = np.zeros(n_rows, bool)
    for s0 in rng.integers(0, n_rows - 5000, 4):
        blocks[s0:s0 + 4500] = True
    one = np.zeros(n_rows, bool)
    one[12345] = True
    masks = {"random 10 %": rng.random(n_rows) < 0.1, "random 90 %": rng.random(n_rows) < 0.9, "every 10th row": np.arange(n_rows) % 10 == 0, "blocks of rows": blocks,
             "only long rows": deg > 508, "only empty rows": deg == 0, "one row": one, "everything": np.ones(n_rows, bool), "the last rows": np.arange(n_rows) >= n_rows - 3}
    for name, selrows in masks.items():
        mask = np.where(selrows, 2, 0).astype(NP[t])
        for mul, add, init in (("MULT", "PLUS", 3), ("PLUS", "MIN", 1000)):
            want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, "NQZERO", init)
            for attempt in (0, 1):
                r = torch.full((n_rows,), 77, dtype=kernels.torch_dtype(t), device="cuda")
                kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dev(mask), dv, r, mul, add, "NQZERO", init)
                torch.cuda.synchronize()
            probe = plan.mask_probe()
            assert probe["served_by_rows"], (name, probe)
            got = host(r, NP[t])
            bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
            assert len(bad) == 0, (t, name, mul, add, len(bad), int(bad[0]), int(deg[bad[0]]), got[bad[0]], want[bad[0]])
    plan.close()


@pytest.mark.gpu
def test_early_exit_over_long_rows_is_the_sequential_fold_signs_of_zeros_included(oracle):
    """early_exit = true (src/cpu/cpu_mxv.hpp:96-98: a row stops behind the first product that moves its sum off init) over rows of up to 20 000 entries: since round 6
    the wavefront searches a long row 64 entries a step instead of one lane walking it.  The search is the sequential fold exactly — also where the sum's BITS change
    without the sum moving off init (a zero whose sign flips: SECOND, MULT or DIV as op_add over products that are +0 / -0): rows whose products are zeros of either
    sign with the first non-zero product early, late or never.  fp32 and int32, several ops, init +0 and -0; every form against the oracle bit for bit."""
    from gpu_utils import hip_mxv
    rng = np.random.default_rng(4711)
    n_cols = 30000
    lens = [0, 1, 10, 63, 64, 65, 128, 129, 300, 1000, 1025, 5000, 20000, 7, 640]
    n_rows = len(lens)
    Ap = np.zeros(n_rows + 1, np.uint32)
    Ap[1:] = np.cumsum(lens)
    nnz = int(Ap[-1])
    Aj = np.empty(nnz, np.uint32)
    for i, d in enumerate(lens):
        Aj[Ap[i]:Ap[i + 1]] = np.sort(rng.choice(n_cols, d, replace=False))
    for t in ("FLOAT", "INT"):
        if t == "FLOAT":
            Ax = rng.choice(np.array([1.0, -1.0, 2.0, -0.5], np.float32), nnz)
            v = rng.choice(np.array([0.0, -0.0], np.float32), n_cols)  # every product a zero of either sign ...
        else:
            Ax = rng.integers(-3, 4, nnz).astype(np.int32)
            v = np.zeros(n_cols, np.int32)
        for where in ("never", "late", "early"):
            vv = v.copy()
            if where != "never":
                for i, d in enumerate(lens):  # ... except one column per row: the row's last entry ("late") or its third ("early")
                    if d >= 3:
                        vv[Aj[Ap[i] + (d - 1 if where == "late" else 2)]] = 3
            mask = np.zeros(n_rows, NP[t])
            for (mul, add) in (("MULT", "SECOND"), ("MULT", "MULT"), ("MULT", "PLUS"), ("MULT", "MIN"), ("MULT", "MINUS")):
                for init in ((0.0, -0.0) if t == "FLOAT" else (0,)):
                    init = NP[t](init)
                    want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vv, mul, add, "EQZERO", init, True)
                    got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, vv, mul, add, "EQZERO", init, early_exit=True, junk=9)
                    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
                    assert len(bad) == 0, (t, where, mul, add, init, [(int(b), lens[b], got[b], want[b]) for b in bad[:4]])
