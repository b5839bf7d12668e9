"""Parity of the HIP masked vxm (push and pull forms, and the operator API's automatic choice) against the oracle."""
import numpy as np
import pytest

from helpers import (NP, SELECTS, SEMIRINGS_COMMON, SEMIRINGS_INT, SEMIRINGS_ORDERED, assert_same, random_csr,
                     random_mask, random_values)

pytestmark = pytest.mark.gpu
FP32_RTOL = 1e-5


def _semirings(t):
    s = SEMIRINGS_COMMON + SEMIRINGS_ORDERED
    return s + SEMIRINGS_INT if t != "FLOAT" else s


def _push(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel):
    from gpu_utils import dev, host
    from spla_amd import kernels
    ri, rx = kernels.vxm_push(t, n_rows, n_cols, dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(mask, NP[t])),
                              dev(np.asarray(vi, np.uint32)), dev(np.asarray(vx, NP[t])), mul, add, sel)
    return host(ri, np.uint32), host(rx, NP[t])


def _pull(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel, identity=False, sell=False):
    """sell: attach the sliced-ELL copy to the transposed matrix's plan (served by its vxm mode when every index of v is stored)"""
    from gpu_utils import dev, host
    from spla_amd import kernels
    dAp, dAj, dAx = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t]))
    Tp, Tj, Tx = kernels.csr_transpose(n_rows, n_cols, dAp, dAj, dAx)
    plan = kernels.CsrPlan(n_cols, len(Aj), Tp, n_rows, Tj, Tx) if sell is True else kernels.CsrPlan(n_cols, len(Aj), Tp)
    if sell == "cob":        # column-ordered blocks copy of the transposed matrix (its vxm mode serves the identity case)
        plan.attach_cob(n_rows, Tp, Tj, Tx)
    elif sell == "cob-small":  # many ranges / pieces / slabs on a small input
        plan.attach_cob(n_rows, Tp, Tj, Tx, row_bits=8, tile_chunks=2)
    ri, rx = kernels.vxm_pull(plan, t, n_rows, n_cols, Tp, Tj, Tx, dev(np.asarray(mask, NP[t])), dev(np.asarray(vi, np.uint32)),
                              dev(np.asarray(vx, NP[t])), mul, add, sel, identity=identity)
    return host(ri, np.uint32), host(rx, NP[t])


def test_transpose_is_stable_csc():
    import torch
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(2)
    n_rows, n_cols = 900, 700
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 9, "INT", skew=True)
    Ax = np.arange(len(Aj), dtype=np.int32)  # entry ids: lets us check the exact permutation
    Tp, Tj, Tx = kernels.csr_transpose(n_rows, n_cols, dev(Ap), dev(Aj), dev(Ax))
    Tp, Tj, Tx = host(Tp, np.uint32), host(Tj, np.uint32), host(Tx, np.int32)
    rows = np.repeat(np.arange(n_rows), np.diff(Ap.astype(np.int64)))
    order = np.lexsort((rows, Aj))  # by column, then row
    assert np.array_equal(Tx, Ax[order]) and np.array_equal(Tj, rows[order])
    assert np.array_equal(Tp, np.concatenate([[0], np.cumsum(np.bincount(Aj, minlength=n_cols))]))


@pytest.mark.parametrize("form", ["auto", "sort"])
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_push_all_cells_bit_exact(oracle, t, form, monkeypatch):
    """The push form folds every column's products in the reference's order: bit-exact for EVERY op, fp32 PLUS included.  form=auto lets
    small expansions of order-free op_add take the compare-and-swap form (no sort), form=sort forces the expand / sort / fold form."""
    if form == "sort":
        monkeypatch.setenv("SPLAHIP_PUSH_ATOMIC", "0")
    rng = np.random.default_rng(31)
    n_rows, n_cols = 1500, 1100
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 8, t, skew=True, long_rows=2, long_len=900)
    for k, (mul, add) in enumerate(_semirings(t)):
        sel = SELECTS[k % len(SELECTS)]
        mask = random_mask(rng, n_cols, t)
        nv = int(rng.integers(1, n_rows))
        vi = np.sort(rng.permutation(n_rows)[:nv]).astype(np.uint32)
        if k % 3 == 0:
            vi = rng.permutation(vi).astype(np.uint32)  # caller-ordered COO
        vx = random_values(rng, nv, t, zeros=0.2)
        want = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
        got = _push(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
        np.testing.assert_array_equal(got[0], want[0], err_msg=f"{t} {mul}/{add} {sel} idx")
        np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32), err_msg=f"{t} {mul}/{add} {sel} val")


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_pull_commutative_cells(oracle, t):
    rng = np.random.default_rng(37)
    n_rows, n_cols = 1700, 1300
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 9, t, skew=True, long_rows=2, long_len=1200)
    # (fp32 MULT as op_add is not reorderable — a tree of products overflows where the left fold does not — and is refused by the pull form)
    sems = [s for s in SEMIRINGS_COMMON + (SEMIRINGS_INT if t != "FLOAT" else []) if not (t == "FLOAT" and s[1] == "MULT")]
    for k, (mul, add) in enumerate(sems):
        sel = SELECTS[k % len(SELECTS)]
        mask = random_mask(rng, n_cols, t)
        nv = n_rows if k % 2 == 0 else int(rng.integers(1, n_rows))
        vi = np.sort(rng.permutation(n_rows)[:nv]).astype(np.uint32)
        vx = random_values(rng, nv, t, zeros=0.2)
        want = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
        for form in ((False, True, "cob", "cob-small") if nv == n_rows else (False,)):  # the sliced-ELL / column-ordered-blocks vxm modes serve the identity case
            got = _pull(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel, identity=(nv == n_rows), sell=form)
            np.testing.assert_array_equal(got[0], want[0], err_msg=f"{t} {mul}/{add} {sel} idx sell={form}")
            if t == "FLOAT" and add in ("PLUS", "MULT"):
                np.testing.assert_allclose(got[1], want[1], rtol=FP32_RTOL, err_msg=f"{t} {mul}/{add} sell={form}")
            else:
                np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32), err_msg=f"{t} {mul}/{add} {sel} val sell={form}")


def test_vxm_naive_reference_case_through_api():
    # reference tests/test_vxm.cpp:33-89: the mxv numbers through v x M with M 5x4
    from spla_amd import api
    M_, N_ = 4, 5
    ir, imask, iv = api.Vector(M_, api.INT), api.Vector(M_, api.INT), api.Vector(N_, api.INT)
    iM = api.Matrix((N_, M_), api.INT)
    for i, x in enumerate([1, 0, 1, 0]):
        imask.set(i, x)
    for i, x in enumerate([3, 0, 3, 0, -1]):
        iv.set(i, x)
    for (i, j, x) in [(0, 1, 2), (0, 4, -9), (1, 0, 2), (1, 4, -8), (2, 2, 3), (3, 4, -1)]:
        iM.set(j, i, x)
    iv.vxm(imask, iM, api.INT.MULT, api.INT.PLUS, api.INT.EQZERO, out=ir, init=api.Scalar(api.INT, 0))
    assert [ir.get(i) for i in range(M_)] == [0, 14, 0, 1]


def test_vxm_explicit_zeros_are_kept_through_api():
    # SURVEY.md 8c probe (iv): v=[3,2], M=[[0,5],[.,7]] -> ([0,1],[0,29])
    from spla_amd import api
    M = api.Matrix.from_lists([0, 0, 1], [0, 1, 1], [0, 5, 7], (2, 2), api.INT)
    v = api.Vector.from_lists([0, 1], [3, 2], 2, api.INT)
    mask = api.Vector.from_lists([0, 1], [1, 1], 2, api.INT)
    r = v.vxm(mask, M, api.INT.MULT, api.INT.PLUS, api.INT.NQZERO)
    assert r.to_lists() == ([0, 1], [0, 29])


@pytest.mark.parametrize("t", ["INT", "FLOAT"])
def test_vxm_api_auto_form_vs_oracle(oracle, t):
    """Sparse frontier (push) and full frontier (pull) through spla_Exec_vxm_masked."""
    from spla_amd import api
    T = getattr(api, t)
    rng = np.random.default_rng(41)
    n_rows, n_cols = 2000, 1500
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 10, t, skew=True)
    M = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), T)
    for nv, (mul, add, sel) in [(30, ("MULT", "PLUS", "ALWAYS")), (n_rows, ("MULT", "PLUS", "NQZERO")), (n_rows // 2, ("PLUS", "MIN", "EQZERO")),
                                (n_rows, ("MULT", "SECOND", "ALWAYS"))]:
        mask_np = random_mask(rng, n_cols, t)
        vi = np.sort(rng.permutation(n_rows)[:nv]).astype(np.uint32)
        vx = random_values(rng, nv, t)
        mask = api.Vector.from_lists(np.arange(n_cols), mask_np, n_cols, T)
        v = api.Vector.from_lists(vi, vx, n_rows, T)
        r = v.vxm(mask, M, getattr(T, mul), getattr(T, add), getattr(T, sel))
        I, V = r.read()
        want = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask_np, vi, vx, mul, add, sel)
        np.testing.assert_array_equal(I, want[0])
        assert_same(V, want[1], t, what=f"{t} nv={nv} {mul}/{add}")


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_push_small_frontier_form(oracle, t):
    """splahip_vxm_push_small: probe + compare-and-swap expansion + compaction with one host synchronisation.  Order-free op_add cells
    against the oracle over many consecutive calls (the accumulator workspace must come back all-zero every time), unsorted frontiers,
    explicit zero products, and the hand-over when the device-side product count exceeds the limit."""
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(77)
    n_rows, n_cols = 2500, 1900
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 8, t, skew=True, long_rows=2, long_len=1200)
    dAp, dAj, dAx = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t]))
    cells = [("MULT", "MIN"), ("PLUS", "MAX"), ("FIRST", "MIN"), ("SECOND", "MAX")]
    if t != "FLOAT":
        cells += [("BAND", "BOR"), ("MULT", "PLUS"), ("MULT", "BXOR"), ("LAND", "LOR")]
    for rep in range(3):
        for mul, add in cells:
            for sel in ("ALWAYS", "EQZERO"):
                mask = random_mask(rng, n_cols, t)
                nv = int(rng.integers(1, 60))
                vi = rng.permutation(n_rows)[:nv].astype(np.uint32)  # unsorted: an order-free fold does not care
                vx = random_values(rng, nv, t, zeros=0.3)
                want_i, want_x = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
                done, products, srt, ri, rx = kernels.vxm_push_small(t, n_rows, n_cols, dAp, dAj, dAx, dev(np.asarray(mask, NP[t])), dev(vi),
                                                                     dev(np.asarray(vx, NP[t])), mul, add, sel, 1 << 20)
                assert done and products == int(np.diff(Ap.astype(np.int64))[vi].sum())
                assert srt == bool(np.all(np.diff(vi.astype(np.int64)) > 0))
                assert np.array_equal(host(ri, np.uint32), want_i), (mul, add, sel)
                assert np.array_equal(host(rx, NP[t]).view(np.uint32), want_x.view(np.uint32)), (mul, add, sel)
    # over the limit: nothing computed, the probe's answer handed back
    vi = np.arange(0, n_rows, 2, dtype=np.uint32)
    vx = random_values(rng, len(vi), t)
    done, products, srt, ri, rx = kernels.vxm_push_small(t, n_rows, n_cols, dAp, dAj, dAx, dev(np.zeros(n_cols, NP[t])), dev(vi), dev(np.asarray(vx, NP[t])),
                                                         "MULT", "MIN", "ALWAYS", 100)
    assert not done and len(ri) == 0 and srt and products == int(np.diff(Ap.astype(np.int64))[vi].sum())
    # ... and the next small call still starts from a clean workspace
    vi = np.array([5], np.uint32)
    vx = random_values(rng, 1, t)
    want_i, want_x = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, np.zeros(n_cols, NP[t]), vi, vx, "MULT", "MIN", "ALWAYS")
    done, _, _, ri, rx = kernels.vxm_push_small(t, n_rows, n_cols, dAp, dAj, dAx, dev(np.zeros(n_cols, NP[t])), dev(vi), dev(np.asarray(vx, NP[t])), "MULT", "MIN",
                                                "ALWAYS", 1 << 20)
    assert done and np.array_equal(host(ri, np.uint32), want_i) and np.array_equal(host(rx, NP[t]).view(np.uint32), want_x.view(np.uint32))


@pytest.mark.parametrize("add", ["MIN", "MAX"])
def test_fp32_min_max_nan_products_every_form(oracle, add, monkeypatch):
    """fp32 MIN / MAX with NaN among the products (reference src/op.cpp:151-155 via src/cpu/cpu_vxm.hpp:101-107): a column folds from its FIRST
    product, so a NaN there is the result, and a NaN anywhere else is never picked.  Every form of vxm — the compare-and-swap push (must notice the
    NaN and hand over to the ordered form), the sort form with its wave-folded long runs, the pull kernels over the CSR of the transpose, its
    sliced-ELL copy and its column-ordered blocks — must reproduce that bit for bit."""
    rng = np.random.default_rng(97 + (add == "MAX"))
    t = "FLOAT"
    n_rows, n_cols = 1700, 1300
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 9, t, skew=True, long_rows=3, long_len=1200)
    nnz = len(Aj)
    Ax = (rng.integers(1, 400, nnz) / 8.0).astype(np.float32)
    for frac_nan, nv in ((0.05, n_rows), (0.3, n_rows), (0.1, 600), (0.5, 40)):
        vi = np.sort(rng.permutation(n_rows)[:nv]).astype(np.uint32)
        vx = (rng.integers(-500, 5000, nv) / 4.0).astype(np.float32)
        vx[rng.random(nv) < 0.2] = np.inf
        vx[rng.random(nv) < frac_nan] = np.nan
        if nv == n_rows:
            vx[0] = np.nan  # the first product of every column row 0 reaches
        for sel in ("ALWAYS", "NQZERO"):
            mask = random_mask(rng, n_cols, t)
            want = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, "PLUS", add, sel)
            assert np.isnan(want[1]).any() and not np.isnan(want[1]).all()
            for form in ("auto", "sort"):
                monkeypatch.setenv("SPLAHIP_PUSH_ATOMIC", "0" if form == "sort" else "1")
                got = _push(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, "PLUS", add, sel)
                np.testing.assert_array_equal(got[0], want[0], err_msg=f"push {form} idx")
                np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32), err_msg=f"push {form} val nv={nv}")
            monkeypatch.delenv("SPLAHIP_PUSH_ATOMIC")
            for form in ((False, True, "cob", "cob-small") if nv == n_rows else (False,)):
                got = _pull(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, "PLUS", add, sel, identity=(nv == n_rows), sell=form)
                np.testing.assert_array_equal(got[0], want[0], err_msg=f"pull idx sell={form}")
                np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32), err_msg=f"pull val sell={form} nv={nv}")


@pytest.mark.parametrize("t,mul,add,xval", [("INT", "BAND", "BOR", 1), ("UINT", "MULT", "MAX", 3), ("INT", "LAND", "BOR", 7)])
def test_pull_first_hit_form_of_a_traversal_level(oracle, t, mul, add, xval):
    """The pull form of a traversal's level (reference src/algorithm.cpp:93-99) settles an output entry at its first stored in-neighbour (mxv_first_hit.hpp, VXM
    mode) when the transposed matrix's owner said its values are one word and the caller's product count says the frontier's rows hold much of the matrix; the
    launch checks the frontier's stored values itself — several words: the complete fold.  Same index-sorted COO as the reference in every case, explicit zeros
    among the stored values included (they are products like any other)."""
    from gpu_utils import dev, host
    from spla_amd import kernels
    rng = np.random.default_rng(4711)
    n_rows, n_cols = 3000, 2600
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 10, "INT", skew=True, long_rows=4, long_len=2000)
    Ax = np.full(len(Aj), xval, NP[t])
    dAp, dAj, dAx = dev(Ap), dev(Aj), dev(Ax)
    Tp, Tj, Tx = kernels.csr_transpose(n_rows, n_cols, dAp, dAj, dAx)
    plan = kernels.CsrPlan(n_cols, len(Aj), Tp)
    plan.set_uniform_values(True, xval)
    deg = np.diff(Ap.astype(np.int64))
    for dens, values in ((0.5, "one"), (0.5, "zero"), (0.5, "mixed"), (0.02, "one"), (1.0, "one")):
        vi = np.nonzero(rng.random(n_rows) < dens)[0].astype(np.uint32)
        vx = {"one": np.full(len(vi), 2, NP[t]), "zero": np.zeros(len(vi), NP[t]), "mixed": rng.integers(0, 5, len(vi)).astype(NP[t])}[values]
        for sel in ("EQZERO", "ALWAYS"):
            mask = (rng.random(n_cols) < 0.3).astype(NP[t])
            wi, wx = oracle.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
            for hint in (int(deg[vi].sum()), 0):  # with the probe's count (first hit where the frontier is heavy enough) and without (the wave-tile kernels)
                plan.hint_products(hint)
                ri, rx = kernels.vxm_pull(plan, t, n_rows, n_cols, Tp, Tj, Tx, dev(mask), dev(vi), dev(vx), mul, add, sel, identity=len(vi) == n_rows)
                assert np.array_equal(host(ri, np.uint32), wi) and np.array_equal(host(rx, NP[t]), wx), (dens, values, sel, hint)
    plan.close()
