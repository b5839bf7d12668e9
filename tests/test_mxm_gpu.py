"""Parity of the HIP SpGEMM operators (mxm: ESC; mxmT_masked: per-mask-entry intersection) against the oracle."""
import numpy as np
import pytest

from helpers import NP, assert_same, random_csr, random_mask, random_values

pytestmark = pytest.mark.gpu
FP32_RTOL = 1e-5


def _dev3(A, t):
    from gpu_utils import dev
    return dev(A[0]), dev(A[1]), dev(np.asarray(A[2], NP[t]))


def _host3(R, t):
    from gpu_utils import host
    return host(R[0], np.uint32), host(R[1], np.uint32), host(R[2], NP[t])


def _same_csr(got, want, t, exact, what):
    np.testing.assert_array_equal(got[0], want[0], err_msg=what + " Rp")
    np.testing.assert_array_equal(got[1], want[1], err_msg=what + " Rj")
    if exact:
        np.testing.assert_array_equal(got[2].view(np.uint32), want[2].view(np.uint32), err_msg=what + " Rx")
    else:
        np.testing.assert_allclose(got[2], want[2], rtol=FP32_RTOL, err_msg=what + " Rx")


CASES = [("MULT", "PLUS", 0), ("PLUS", "MIN", 100), ("MULT", "SECOND", 0), ("MIN", "MAX", 0), ("MULT", "MINUS", 0), ("LAND", "LOR", 0)]


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxm_vs_oracle(oracle, t):
    from spla_amd import kernels
    rng = np.random.default_rng(51)
    m, k, n = 400, 350, 300
    A = random_csr(rng, m, k, 6, t, skew=True, long_rows=1, long_len=300)
    B = random_csr(rng, k, n, 6, t, skew=True, long_rows=1, long_len=280)
    for mul, add, init in CASES:
        want = oracle.mxm(t, m, n, A, B, mul, add, init)
        for block in (0, 2000):  # one row block, and many small row blocks
            got, products = kernels.mxm(t, m, n, _dev3(A, t), _dev3(B, t), mul, add, init, block_products=block)
            # runs up to 64 products are folded strictly in order (bit-exact, fp32 PLUS too); longer runs of an
            # associative op use an in-order tree -> fp32 PLUS/MULT within 1e-5
            exact = not (t == "FLOAT" and add in ("PLUS", "MULT"))
            _same_csr(_host3(got, t), want, t, exact, f"mxm {t} {mul}/{add} block={block}")
            assert products == sum(int(B[0][c + 1]) - int(B[0][c]) for c in A[1])


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxmT_masked_vs_oracle(oracle, t):
    from spla_amd import kernels
    rng = np.random.default_rng(53)
    m, k, n = 500, 600, 450
    A = random_csr(rng, m, k, 12, t, skew=True, long_rows=3, long_len=400)
    B = random_csr(rng, n, k, 12, t, skew=True, long_rows=3, long_len=420)
    Mk = random_csr(rng, m, n, 20, t)
    Mk = (Mk[0], Mk[1], random_mask(rng, len(Mk[1]), t))
    for (mul, add, init), sel in zip(CASES, ["GTZERO", "ALWAYS", "EQZERO", "NQZERO", "ALWAYS", "GEZERO"]):
        want = oracle.mxmT_masked(t, m, Mk, A, B, mul, add, sel, init)
        got = kernels.mxmT_masked(t, m, _dev3(Mk, t), _dev3(A, t), _dev3(B, t), mul, add, sel, init)
        exact = not (t == "FLOAT" and add in ("PLUS", "MULT"))
        _same_csr(_host3(got, t), want, t, exact, f"mxmT {t} {mul}/{add} {sel}")


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxmT_row_form_vs_oracle(oracle, t, monkeypatch):
    """The opt-in row form of the heavy mask entries (SPLAHIP_MXMT_ROW=1: A[i, :] in a wavefront's LDS with a one-hash filter, B[j, :]
    streamed against it — mxm.hip: mxmT_row_kernel).  Rows on both sides of its two LDS layouts (<= 1023 / <= 4095 entries), rows it leaves
    to the per-entry kernel (> 4095), B rows far longer than the A row (searched, not streamed), and an op_add it must refuse
    (FIRST is not commutative)."""
    from spla_amd import kernels
    monkeypatch.setenv("SPLAHIP_MXMT_ROW", "1")
    rng = np.random.default_rng(59)
    m, k, n = 260, 9000, 300
    A = random_csr(rng, m, k, 40, t, skew=True, long_rows=6, long_len=900)
    A2 = random_csr(rng, m, k, 40, t, skew=True, long_rows=4, long_len=3000)
    A3 = random_csr(rng, m, k, 40, t, skew=True, long_rows=2, long_len=5000)
    B = random_csr(rng, n, k, 60, t, skew=True, long_rows=8, long_len=7000)
    Mk = random_csr(rng, m, n, 120, t)
    Mk = (Mk[0], Mk[1], random_mask(rng, len(Mk[1]), t))
    for Am in (A, A2, A3):
        for (mul, add, init), sel in zip(CASES + [("MULT", "FIRST", 0)], ["GTZERO", "ALWAYS", "EQZERO", "NQZERO", "ALWAYS", "GEZERO", "ALWAYS"]):
            want = oracle.mxmT_masked(t, m, Mk, Am, B, mul, add, sel, init)
            got = kernels.mxmT_masked(t, m, _dev3(Mk, t), _dev3(Am, t), _dev3(B, t), mul, add, sel, init)
            exact = not (t == "FLOAT" and add in ("PLUS", "MULT"))
            _same_csr(_host3(got, t), want, t, exact, f"mxmT row form {t} {mul}/{add} {sel}")


def test_mxm_naive_reference_case():
    # reference tests/test_mxm.cpp:33-94: 3x4 . 4x2 fp32 known answer
    from spla_amd import api
    A = api.Matrix((3, 4), api.FLOAT)
    B = api.Matrix((4, 2), api.FLOAT)
    for (i, j, x) in [(0, 0, 1), (0, 1, 2), (0, 3, 3), (1, 1, 4), (1, 2, 5), (2, 0, 6), (2, 3, 7)]:
        A.set(i, j, float(x))
    for (i, j, x) in [(0, 0, 1), (0, 1, 2), (1, 0, 3), (2, 1, 4), (3, 0, 5), (3, 1, 6)]:
        B.set(i, j, float(x))
    R = A.mxm(B, api.FLOAT.MULT, api.FLOAT.PLUS, init=api.Scalar(api.FLOAT, 0.0))
    dense = np.zeros((3, 2), np.float32)
    for (i, j, x) in [(0, 0, 1), (0, 1, 2), (0, 3, 3), (1, 1, 4), (1, 2, 5), (2, 0, 6), (2, 3, 7)]:
        for (p, q, y) in [(0, 0, 1), (0, 1, 2), (1, 0, 3), (2, 1, 4), (3, 0, 5), (3, 1, 6)]:
            if j == p:
                dense[i, q] += x * y
    for i in range(3):
        for j in range(2):
            assert R.get(i, j) == dense[i, j]
    I, J, V = R.read()  # Matrix::read() works on results here (the reference crashes)
    assert len(I) == np.count_nonzero(dense)


def test_mxm_cancelled_entry_is_dropped():
    # SURVEY.md 8c probe (v): [1,-1] . [4,4]^T -> empty
    from spla_amd import api
    A = api.Matrix.from_lists([0, 0], [0, 1], [1.0, -1.0], (1, 2), api.FLOAT)
    B = api.Matrix.from_lists([0, 1], [0, 0], [4.0, 4.0], (2, 1), api.FLOAT)
    R = A.mxm(B, api.FLOAT.MULT, api.FLOAT.PLUS)
    assert R.to_lists() == ([], [], [])


def test_mxmT_naive_reference_case():
    # reference tests/test_mxmT.cpp:33-103: 3x4 . (2x4)^T fp32 with a GTZERO mask -> [[23,0],[0,0],[83,0]]
    from spla_amd import api
    R = api.Matrix((3, 2), api.FLOAT)
    mask = api.Matrix((3, 2), api.FLOAT)
    A = api.Matrix((3, 4), api.FLOAT)
    B = api.Matrix((2, 4), api.FLOAT)
    for (i, j) in [(0, 0), (0, 1), (2, 0), (2, 1)]:
        mask.set(i, j, 1.0)
    for (i, j, x) in [(0, 0, 1.0), (0, 2, 2.0), (1, 1, 3.0), (1, 3, 4.0), (2, 0, 5.0), (2, 2, 6.0)]:
        A.set(i, j, x)
    for (i, j, x) in [(0, 0, 7.0), (0, 2, 8.0), (1, 1, 9.0), (1, 3, 10.0)]:
        B.set(i, j, x)
    A.mxmT(mask, B, api.FLOAT.MULT, api.FLOAT.PLUS, api.FLOAT.GTZERO, out=R, init=api.Scalar(api.FLOAT, 0.0))
    assert [[R.get(i, j) for j in range(2)] for i in range(3)] == [[23.0, 0.0], [0.0, 0.0], [83.0, 0.0]]
    # entries equal to init are dropped from the structure (CPU backend behaviour)
    assert R.to_lists() == ([0, 2], [0, 0], [23.0, 83.0])


def test_triangle_count_form():
    # reference src/algorithm.cpp:398-399: B = (L x L^T) .L with MULT/PLUS/GTZERO, triangles = sum of B
    from spla_amd import api
    rng = np.random.default_rng(61)
    n = 300
    dense = np.triu(rng.random((n, n)) < 0.05, 1)
    dense = dense | dense.T
    I, J = np.nonzero(np.tril(dense, -1))
    L = api.Matrix.from_lists(I, J, np.ones(len(I), np.int32), (n, n), api.INT)
    Bm = L.mxmT(L, L, api.INT.MULT, api.INT.PLUS, api.INT.GTZERO)
    _, _, V = Bm.read()
    a = dense.astype(np.int64)
    assert int(V.sum()) == int(np.trace(a @ a @ a) // 6)


def test_mxm_refuses_a_result_beyond_32bit_row_pointers(monkeypatch):
    """The entry count of the product is summed in 64 bits and checked before anything is sized from it (reference row pointers are uint,
    include/spla/config.hpp:56); SPLAHIP_MXM_MAX_NNZ lowers the limit so the guard can be exercised on a small input: the call must fail with
    NOT_IMPLEMENTED, leave nothing behind, and the same call without the limit must still work."""
    from spla_amd import _lib, kernels
    rng = np.random.default_rng(5)
    A = random_csr(rng, 300, 300, 8, "INT")
    dA = _dev3(A, "INT")
    ok, _ = kernels.mxm("INT", 300, 300, dA, dA, "MULT", "PLUS", 0, block_products=500)
    nnz = int(ok[1].numel())
    assert nnz > 100
    monkeypatch.setenv("SPLAHIP_MXM_MAX_NNZ", str(nnz // 3))
    with pytest.raises(_lib.SplaHipError) as err:
        kernels.mxm("INT", 300, 300, dA, dA, "MULT", "PLUS", 0, block_products=500)  # many row blocks: the limit is crossed mid-way
    assert err.value.status == 1024 and "32-bit row pointers" in str(err.value)
    monkeypatch.delenv("SPLAHIP_MXM_MAX_NNZ")
    again, _ = kernels.mxm("INT", 300, 300, dA, dA, "MULT", "PLUS", 0, block_products=500)
    assert int(again[1].numel()) == nnz


def _last_info():
    import ctypes as C
    from spla_amd import _lib
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _lib.call("splahip_mxm_last_info", C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxm_row_classes_heavy_rows_vs_oracle(oracle, t):
    """Rows of more than 4096 products take the LDS-accumulator form (mxm_spa.hip): several column blocks (30,000 columns), chunks of more than
    1024 entries per row, rows mixed with light ones (ESC), cancellations that fold to init (dropped, then compacted), every op_add with an LDS
    atomic; the ordered / non-atomic op_add cases of the same input stay with ESC and stay exact."""
    from spla_amd import kernels
    rng = np.random.default_rng(61)
    m, k, n = 700, 2600, 30000
    A = random_csr(rng, m, k, 45, t, skew=True, long_rows=3, long_len=2500, empty_frac=0.1)
    B = random_csr(rng, k, n, 55, t, skew=True, long_rows=2, long_len=20000)
    dA, dB = _dev3(A, t), _dev3(B, t)
    cases = [("MULT", "PLUS", 0), ("PLUS", "MIN", 100), ("LAND", "LOR", 0), ("MULT", "LAND", 1), ("MIN", "MAX", 0), ("MULT", "SECOND", 0)]
    if t != "FLOAT":
        cases += [("BAND", "BOR", 0), ("MULT", "BXOR", 0), ("BOR", "BAND", 0)]
    for mul, add, init in cases:
        want = oracle.mxm(t, m, n, A, B, mul, add, init)
        got, products = kernels.mxm(t, m, n, dA, dB, mul, add, init)
        heavy, light_nnz, dense_h = _last_info()
        served = not (add == "SECOND" or (t == "FLOAT" and add in ("MIN", "MAX")))
        assert (heavy > 0) == served, (mul, add, heavy)
        exact = not (t == "FLOAT" and add in ("PLUS", "MULT"))
        _same_csr(_host3(got, t), want, t, exact, f"mxm row classes {t} {mul}/{add}")
        assert products == sum(int(B[0][c + 1]) - int(B[0][c]) for c in A[1])


def test_mxm_dense_corner_on_the_matrix_cores(oracle, monkeypatch):
    """SURVEY / north star g1: where a row bucket degenerates to a dense tile the product runs on MFMA.  A and B carry a dense 160 x 160 leading
    block (90 % full) inside sparse matrices: splahip_mxm densifies the corners, multiplies them with v_mfma_f32_32x32x2_f32 (values and product
    counts), pre-loads the accumulators of rows < h with the tile and lets the sparse traversal skip those products.  Same structure as the oracle,
    values within 1e-5; identical structure and 1e-5 agreement with the all-sparse run (SPLAHIP_MXM_DENSE=0)."""
    from spla_amd import kernels
    rng = np.random.default_rng(67)
    n = 1200
    def with_block(seed):
        r = np.random.default_rng(seed)
        Ap, Aj, Ax = random_csr(r, n, n, 30, "FLOAT", skew=True)
        import scipy.sparse as sp
        S = sp.csr_matrix((Ax, Aj.astype(np.int64), Ap.astype(np.int64)), shape=(n, n)).tolil()
        blk = (r.random((160, 160)) < 0.9) * (r.integers(1, 64, (160, 160)) / 16.0)
        S[:160, :160] = blk.astype(np.float32)
        S = S.tocsr()
        S.sort_indices()
        S.eliminate_zeros()
        return S.indptr.astype(np.uint32), S.indices.astype(np.uint32), S.data.astype(np.float32)
    A, B = with_block(1), with_block(2)
    want = oracle.mxm("FLOAT", n, n, A, B, "MULT", "PLUS", 0.0)
    dA, dB = _dev3(A, "FLOAT"), _dev3(B, "FLOAT")
    got, _ = kernels.mxm("FLOAT", n, n, dA, dB, "MULT", "PLUS", 0.0)
    heavy, _, dense_h = _last_info()
    assert heavy > 0 and dense_h >= 128, (heavy, dense_h)
    _same_csr(_host3(got, "FLOAT"), want, "FLOAT", False, "mxm dense corner (MFMA)")
    monkeypatch.setenv("SPLAHIP_MXM_DENSE", "0")
    got2, _ = kernels.mxm("FLOAT", n, n, dA, dB, "MULT", "PLUS", 0.0)
    assert _last_info()[2] == 0
    a, b = _host3(got, "FLOAT"), _host3(got2, "FLOAT")
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-5, atol=0)


def _last_dense():
    import ctypes as C
    from spla_amd import _lib
    w = (C.c_uint32 * 6)()
    _lib.call("splahip_mxm_last_dense", w)
    return tuple(int(x) for x in w)


@pytest.mark.gpu
@pytest.mark.parametrize("i0,k0,j0,hi,hk,hj,n", [(352, 608, 96, 160, 128, 192, 1500), (64, 0, 2016, 128, 128, 128, 2600), (0, 416, 0, 96, 160, 96, 900)])
def test_mxm_dense_window_away_from_the_origin(oracle, monkeypatch, i0, k0, j0, hi, hk, hj, n):
    """north star: "MFMA ... where a row-bucket degenerates to a dense tile" — ANYWHERE, not only in the leading corner.  A carries a dense hi x hk block at
    (i0, k0), B a dense hk x hj block at (k0, j0) (90 % full, rectangular, off the diagonal; the second case's columns straddle a 2048-column accumulator
    block), both inside sparse skewed matrices.  splahip_mxm finds the half-full 32 x 32 tiles (tile_fill_kernel), grows and verifies the window, multiplies
    the blocks with the LDS-staged v_mfma_f32_32x32x2_f32 loop (values and product counts), pre-loads the window's cells of the rows it meets and leaves the
    window's products out of the sparse walk.  Same structure as the oracle, values within 1e-5 (reference: src/cpu/cpu_mxm.hpp:85-114); identical structure
    and 1e-5 agreement with the all-sparse run (SPLAHIP_MXM_DENSE=0)."""
    from spla_amd import kernels
    import scipy.sparse as sp

    def with_block(seed, r0, c0, hr, hc):
        r = np.random.default_rng(seed)
        Ap, Aj, Ax = random_csr(r, n, n, 12, "FLOAT", skew=True)
        S = sp.csr_matrix((Ax, Aj.astype(np.int64), Ap.astype(np.int64)), shape=(n, n)).tolil()
        blk = (r.random((hr, hc)) < 0.9) * (r.integers(1, 64, (hr, hc)) / 16.0)
        S[r0:r0 + hr, c0:c0 + hc] = blk.astype(np.float32)
        S = S.tocsr()
        S.sort_indices()
        S.eliminate_zeros()
        return S.indptr.astype(np.uint32), S.indices.astype(np.uint32), S.data.astype(np.float32)

    A, B = with_block(11, i0, k0, hi, hk), with_block(12, k0, j0, hk, hj)
    want = oracle.mxm("FLOAT", n, n, A, B, "MULT", "PLUS", 0.0)
    dA, dB = _dev3(A, "FLOAT"), _dev3(B, "FLOAT")
    got, _ = kernels.mxm("FLOAT", n, n, dA, dB, "MULT", "PLUS", 0.0)
    heavy, _, dense_rows = _last_info()
    w = _last_dense()
    assert heavy > 0 and dense_rows >= 64, (heavy, dense_rows, w)
    # the window lies inside the planted blocks (tiles are 32-aligned: a block that does not start on a multiple of 32 loses its ragged edge) and away from the origin
    assert w[3] >= 64 and w[4] >= 64 and w[5] >= 64, w
    assert i0 <= w[0] and w[0] + w[3] <= i0 + hi + 31 and k0 <= w[1] and w[1] + w[4] <= k0 + hk + 31 and j0 <= w[2] and w[2] + w[5] <= j0 + hj + 31, w
    assert (w[0], w[1], w[2]) != (0, 0, 0)
    _same_csr(_host3(got, "FLOAT"), want, "FLOAT", False, f"mxm dense window at {w}")
    monkeypatch.setenv("SPLAHIP_MXM_DENSE", "0")
    got2, _ = kernels.mxm("FLOAT", n, n, dA, dB, "MULT", "PLUS", 0.0)
    assert _last_info()[2] == 0 and _last_dense()[3] == 0
    a, b = _host3(got, "FLOAT"), _host3(got2, "FLOAT")
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-5, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxmT_count_mode_of_uniform_values(oracle, t):
    """mxmT_masked(L, L, L) with every stored value one word — the triangle-count form — lets the block merge compute an entry from the NUMBER
    of matches (mxm.hip: count_mode / merge_counts): PLUS and BXOR count, MIN / MAX / BOR / BAND / FIRST / SECOND are idempotent, the
    logical ops and a float sum keep the match list.  Values other than 1, rows long enough for the heavy classes, and one odd value that
    must switch the shortcut off."""
    from spla_amd import kernels
    rng = np.random.default_rng(97)
    n = 700
    dense = np.tril(rng.random((n, n)) < 0.35, -1)
    dense[:, :40] |= np.tril(rng.random((n, 40)) < 0.9, -1)[:, :40]
    I, J = np.nonzero(dense)
    order = np.lexsort((J, I))
    I, J = I[order], J[order]
    Lp = np.zeros(n + 1, np.uint32)
    np.add.at(Lp, I + 1, 1)
    Lp = np.cumsum(Lp).astype(np.uint32)
    Lj = J.astype(np.uint32)
    ops = [("MULT", "PLUS", 0), ("MULT", "MIN", 50), ("PLUS", "MAX", 0), ("MULT", "FIRST", 0), ("MULT", "SECOND", 0), ("LAND", "LOR", 0)]
    if t != "FLOAT":
        ops += [("MULT", "BXOR", 0), ("BAND", "BOR", 1), ("BOR", "BAND", 7)]
    for xval, odd in ((1, False), (3, False), (3, True)):
        Lx = np.full(len(Lj), xval, NP[t])
        if odd:
            Lx[len(Lx) // 3] = 5
        L = (Lp, Lj, Lx)
        dL = _dev3(L, t)
        for mul, add, init in ops:
            want = oracle.mxmT_masked(t, n, L, L, L, mul, add, "ALWAYS", init)
            got = kernels.mxmT_masked(t, n, dL, dL, dL, mul, add, "ALWAYS", init)
            exact = not (t == "FLOAT" and add in ("PLUS", "MULT"))
            _same_csr(_host3(got, t), want, t, exact, f"mxmT count mode {t} x={xval} odd={odd} {mul}/{add}")


@pytest.mark.parametrize("t", ["INT", "UINT"])
@pytest.mark.parametrize("shape", ["M=A=B", "B=A", "B=M"])
def test_mxmT_owner_form_of_the_heavy_entries(oracle, monkeypatch, t, shape):
    """Count mode with heavy entries takes the OWNER form (mxm.hip: mxmT_owner_kernel): an entry belongs to the longer of its two rows, whose columns become a bitmap
    in LDS, and only the shorter row is streamed.  Operands that are the same matrix (the triangle count), B = A under a separate mask, and B = the mask;
    rectangular, empty rows, an inner dimension that is no multiple of 32, masks that reject entries, every op_add that may count matches; bit-exact against the
    oracle and identical to the block-merge form (SPLAHIP_MXMT_OWNER=0).  Reference: src/cpu/cpu_mxmT_masked.hpp:87-125."""
    from spla_amd import kernels
    rng = np.random.default_rng(1234 + len(shape))
    n, k = (900, 900) if shape == "M=A=B" else (700, 1237)

    def struct(rows, cols, p, hot):
        d = rng.random((rows, cols)) < p
        d[:, :hot] |= rng.random((rows, hot)) < 0.8      # hub columns: long rows on both sides
        d[rng.integers(0, rows, 12), :] = False          # a few empty rows
        d[rng.integers(0, rows, 6), :] |= rng.random((6, cols)) < 0.7   # a few very long ones
        I, J = np.nonzero(d)
        P = np.zeros(rows + 1, np.uint32)
        np.add.at(P, I + 1, 1)
        return np.cumsum(P).astype(np.uint32), J.astype(np.uint32)

    xa = 3
    if shape == "M=A=B":
        Ap, Aj = struct(n, n, 0.25, 60)
        A = (Ap, Aj, np.full(len(Aj), xa, NP[t]))
        M = B = A
        rows_m = n
    elif shape == "B=A":
        Ap, Aj = struct(n, k, 0.2, 50)
        A = B = (Ap, Aj, np.full(len(Aj), xa, NP[t]))
        Mp, Mj = struct(n, n, 0.05, 10)
        M = (Mp, Mj, random_mask(rng, len(Mj), t))
        rows_m = n
    else:  # B = M: the mask is n x n and doubles as B (n rows of n columns), A is n x n
        Mp, Mj = struct(n, n, 0.2, 50)
        M = B = (Mp, Mj, np.full(len(Mj), 2, NP[t]))
        Ap, Aj = struct(n, n, 0.2, 40)
        A = (Ap, Aj, np.full(len(Aj), xa, NP[t]))
        rows_m = n
    dA = _dev3(A, t)
    dB = dA if B is A else (_dev3(B, t) if B is not M else None)
    dM = dA if M is A else _dev3(M, t)
    if dB is None:
        dB = dM
    for mul, add, sel, init in [("MULT", "PLUS", "ALWAYS", 0), ("MULT", "PLUS", "NQZERO", 5), ("PLUS", "MIN", "ALWAYS", 100), ("MULT", "BXOR", "ALWAYS", 0),
                                ("BAND", "BOR", "GTZERO", 1), ("MULT", "SECOND", "ALWAYS", 0)]:
        if shape == "B=M" and sel != "ALWAYS":
            continue  # (the mask's values are one word there: every select answers alike)
        want = oracle.mxmT_masked(t, rows_m, M, A, B, mul, add, sel, init)
        got = kernels.mxmT_masked(t, rows_m, dM, dA, dB, mul, add, sel, init)
        _same_csr(_host3(got, t), want, t, True, f"mxmT owner form {shape} {t} {mul}/{add}/{sel}")
        monkeypatch.setenv("SPLAHIP_MXMT_OWNER", "0")
        got2 = kernels.mxmT_masked(t, rows_m, dM, dA, dB, mul, add, sel, init)
        monkeypatch.delenv("SPLAHIP_MXMT_OWNER")
        _same_csr(_host3(got2, t), want, t, True, f"mxmT block merge {shape} {t} {mul}/{add}/{sel}")


@pytest.mark.parametrize("add", ["MIN", "MAX"])
def test_fp32_min_max_nan_products_and_nan_init(oracle, add):
    """mxm / mxmT fold from `init` (src/cpu/cpu_mxm.hpp:93-113, cpu_mxmT_masked.hpp:94-123): under fp32 MIN / MAX a NaN product is never picked and
    a NaN init sticks (and `r != init` then keeps every touched entry).  Long runs / heavy entries are folded by whole wavefronts — same words."""
    from spla_amd import kernels
    t = "FLOAT"
    rng = np.random.default_rng(59 + (add == "MAX"))
    m, k, n = 300, 350, 260
    A = random_csr(rng, m, k, 8, t, skew=True, long_rows=2, long_len=300)
    B = random_csr(rng, k, n, 8, t, skew=True, long_rows=2, long_len=250)
    A = (A[0], A[1], (rng.integers(-50, 400, len(A[1])) / 8.0).astype(np.float32))
    bx = (rng.integers(-50, 400, len(B[1])) / 8.0).astype(np.float32)
    bx[rng.random(len(bx)) < 0.05] = np.nan
    bx[rng.random(len(bx)) < 0.05] = np.inf
    B = (B[0], B[1], bx)
    for init in (np.float32(100.0), np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan)):
        want = oracle.mxm(t, m, n, A, B, "PLUS", add, init)
        got, _ = kernels.mxm(t, m, n, _dev3(A, t), _dev3(B, t), "PLUS", add, init)
        _same_csr(_host3(got, t), want, t, True, f"mxm PLUS/{add} init={init}")
    # mxmT: B^T rows share columns with A rows
    m, k, n = 400, 500, 380
    A = random_csr(rng, m, k, 14, t, skew=True, long_rows=3, long_len=420)
    Bt = random_csr(rng, n, k, 14, t, skew=True, long_rows=3, long_len=400)
    ax = (rng.integers(-50, 400, len(A[1])) / 8.0).astype(np.float32)
    ax[rng.random(len(ax)) < 0.05] = np.nan
    A = (A[0], A[1], ax)
    Bt = (Bt[0], Bt[1], (rng.integers(-50, 400, len(Bt[1])) / 8.0).astype(np.float32))
    Mk = random_csr(rng, m, n, 25, t)
    Mk = (Mk[0], Mk[1], random_mask(rng, len(Mk[1]), t))
    for init, sel in ((np.float32(100.0), "ALWAYS"), (np.float32(np.nan), "NQZERO"), (np.float32(-np.inf), "GTZERO")):
        want = oracle.mxmT_masked(t, m, Mk, A, Bt, "PLUS", add, sel, init)
        got = kernels.mxmT_masked(t, m, _dev3(Mk, t), _dev3(A, t), _dev3(Bt, t), "PLUS", add, sel, init)
        _same_csr(_host3(got, t), want, t, True, f"mxmT PLUS/{add} init={init} {sel}")
