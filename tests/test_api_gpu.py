"""The operator API end to end on the GPU (spla_Exec_* -> dispatcher -> format validation -> HIP kernels -> read back):
the reference's own unit-test cases, written the way the reference writes them, plus randomised checks vs the oracle."""
import numpy as np
import pytest

from helpers import assert_same, random_csr, random_mask, random_values

pytestmark = pytest.mark.gpu


def _api():
    from spla_amd import api
    return api


def test_accelerator_is_hip():
    api = _api()
    api.set_accelerator(api.ACCELERATOR_HIP)
    assert "gfx950" in api.accelerator_info()


def test_mxv_masked_naive_reference_case():
    # reference tests/test_mxv.cpp:33-89
    api = _api()
    M_, N_ = 4, 5
    ir, imask, iv = api.Vector(M_, api.INT), api.Vector(M_, api.INT), api.Vector(N_, api.INT)
    iM = api.Matrix((M_, N_), api.INT)
    for i, x in enumerate([2, -5, 4, 5]):
        ir.set(i, x)
    for i, x in enumerate([1, 0, 1, 0]):
        imask.set(i, x)
    for i, x in enumerate([3, 0, 3, 0, -1]):
        iv.set(i, x)
    for (i, j, x) in [(0, 1, 2), (0, 4, -9), (1, 0, 2), (1, 4, -8), (2, 2, 3), (3, 4, -1)]:
        iM.set(i, j, x)
    iM.mxv(imask, iv, api.INT.MULT, api.INT.PLUS, api.INT.EQZERO, out=ir, init=api.Scalar(api.INT, 0))
    assert [ir.get(i) for i in range(M_)] == [0, 14, 0, 1]


def test_mxv_pyspla_docstring_case():
    # reference python/pyspla/matrix.py:939-964 example: M = [[1,2],[3,4]]-like known answer through from_lists/to_lists
    api = _api()
    M = api.Matrix.from_lists([0, 1, 2, 2, 3], [1, 2, 0, 3, 2], [1, 2, 3, 4, 5], (4, 4), api.INT)
    mask = api.Vector.dense(4, api.INT, 1)
    v = api.Vector.from_lists([2], [1], 4, api.INT)
    r = M.mxv(mask, v, api.INT.LAND, api.INT.LOR, api.INT.NQZERO)
    assert r.to_lists() == ([1, 3], [1, 1])


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_mxv_api_vs_oracle(oracle, t):
    api = _api()
    T = getattr(api, t)
    rng = np.random.default_rng(21)
    n_rows, n_cols = 700, 500
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 7, t, skew=True, long_rows=1, long_len=450)
    M = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), T)
    for mul, add, sel, init, early in [("MULT", "PLUS", "NQZERO", 0, False), ("PLUS", "MIN", "ALWAYS", 100, False),
                                       ("LAND", "LOR", "EQZERO", 0, True), ("MULT", "SECOND", "GTZERO", 0, False)]:
        mask_np = random_mask(rng, n_rows, t)
        v_np = random_values(rng, n_cols, t, zeros=0.2)
        mask = api.Vector.from_lists(np.arange(n_rows), mask_np, n_rows, T)
        v = api.Vector.from_lists(np.arange(n_cols), v_np, n_cols, T)
        r = M.mxv(mask, v, getattr(T, mul), getattr(T, add), getattr(T, sel), init=api.Scalar(T, init),
                  desc=api.Descriptor(early_exit=early))
        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask_np, v_np, mul, add, sel, init, early)
        # read back sparse (entries equal to the fill value 0 are absent) and densify
        I, V = r.read()
        got = np.zeros(n_rows, dtype=want.dtype)
        got[I] = V
        assert np.all(np.diff(I.astype(np.int64)) > 0), "COO read-back must be index-sorted"
        assert_same(got, want, t, what=f"{t} {mul}/{add} {sel}")


@pytest.mark.parametrize("t", ["INT", "FLOAT"])
def test_mxv_api_builds_sliced_ell_copy_for_a_reused_matrix(oracle, t, monkeypatch):
    """a matrix multiplied again and again gets the sliced-ELL copy (here after 2 calls instead of the default 256): the calls before
    and after the switch agree with the oracle, and a matrix that is modified in between drops the copy with its plan"""
    monkeypatch.setenv("SPLA_HIP_SELL_AFTER", "2")
    api = _api()
    T = getattr(api, t)
    rng = np.random.default_rng(22)
    n_rows, n_cols = 900, 800
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 6, t, skew=True, long_rows=2, long_len=600)
    M = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), T)
    cells = [("MULT", "PLUS", "ALWAYS", 0), ("MULT", "PLUS", "EQZERO", 3), ("PLUS", "MIN", "NQZERO", 1000), ("MULT", "PLUS", "ALWAYS", 0)]
    if t == "INT":
        cells.insert(2, ("BAND", "BOR", "EQZERO", 0))
    for mul, add, sel, init in cells:
        mask_np = random_mask(rng, n_rows, t)
        v_np = random_values(rng, n_cols, t, zeros=0.7)
        mask = api.Vector.from_lists(np.arange(n_rows), mask_np, n_rows, T)
        v = api.Vector.from_lists(np.arange(n_cols), v_np, n_cols, T)
        r = M.mxv(mask, v, getattr(T, mul), getattr(T, add), getattr(T, sel), init=api.Scalar(T, init))
        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask_np, v_np, mul, add, sel, init)
        assert_same(r.to_dense(), want, t, what=f"{t} {mul}/{add} {sel}")


def test_type_mismatch_is_an_error():
    api = _api()
    M = api.Matrix.from_lists([0], [0], [1], (1, 1), api.INT)
    v = api.Vector.from_lists([0], [1.0], 1, api.FLOAT)
    mask = api.Vector.from_lists([0], [1], 1, api.INT)
    with pytest.raises(api.SplaError) as e:
        M.mxv(mask, v, api.INT.MULT, api.INT.PLUS, api.INT.NQZERO)
    assert e.value.status == 1  # Status::Error via the dispatcher's exception handler


def test_mxv_layout_copies_through_the_api(oracle, monkeypatch):
    """Matrix::set_format(AccCsr) is "prepare now" (reference include/spla/matrix.hpp:51-65): plan + both layout copies are built before the first
    product; without it the column-ordered-blocks copy arrives with the second call (rent or buy).  Results identical either way."""
    api = _api()
    rng = np.random.default_rng(77)
    n_rows, n_cols = 5000, 4000
    for t, mul, add, init in (("INT", "PLUS", "MIN", 1 << 20), ("FLOAT", "MULT", "PLUS", 0.0), ("UINT", "BAND", "BOR", 0)):
        T = getattr(api, t)
        Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 12, t, skew=True, long_rows=1, long_len=3000)
        mask_np = random_mask(rng, n_rows, t)
        v_np = random_values(rng, n_cols, t, zeros=0.2)
        want = oracle.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask_np, v_np, mul, add, "EQZERO", init)
        for prepare in (False, True):
            M = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), T)
            if prepare:
                M.set_format(6)  # FormatMatrix::AccCsr (include/spla.hpp; reference include/spla/config.hpp:102-122)
            mask = api.Vector.from_lists(np.arange(n_rows), mask_np, n_rows, T)
            v = api.Vector.from_lists(np.arange(n_cols), v_np, n_cols, T)
            for call in range(3):  # call 1: CSR kernels (or the prepared copies); call 2 builds the copy; call 3 runs on it
                r = M.mxv(mask, v, getattr(T, mul), getattr(T, add), T.EQZERO, init=api.Scalar(T, init))
                ri, rx = r.read()
                got = np.full(n_rows, 0, dtype=want.dtype)
                got[ri] = rx
                assert_same(got, want, t, what=f"{t} {mul}/{add} prepare={prepare} call {call + 1}")  # (entries equal to the fill value 0 are not stored)


def test_deterministic_descriptor_gives_bit_identical_fp32_sums(oracle):
    """Descriptor(deterministic=True) (an addition of this build; include/spla.hpp) sends an fp32 PLUS product to the fixed-order kernels although the
    matrix is prepared (column-ordered blocks attached): repeated calls are bit-identical, equal to what an unprepared matrix's CSR kernels give, and within
    1e-5 of the reference's sequential fold like every other form.  The default call on the same matrix stays within 1e-5 too (its last bit may vary)."""
    api = _api()
    rng = np.random.default_rng(78)
    n_rows, n_cols = 6000, 5000
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 14, "FLOAT", skew=True, long_rows=2, long_len=4000)
    mask_np = random_mask(rng, n_rows, "FLOAT")
    v_np = random_values(rng, n_cols, "FLOAT", zeros=0.1)
    want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask_np, v_np, "MULT", "PLUS", "NQZERO", 0.0)
    T = api.FLOAT

    def product(M, desc):
        mask = api.Vector.from_lists(np.arange(n_rows), mask_np, n_rows, T)
        v = api.Vector.from_lists(np.arange(n_cols), v_np, n_cols, T)
        return M.mxv(mask, v, T.MULT, T.PLUS, T.NQZERO, init=api.Scalar(T, 0.0), desc=desc).to_dense()

    plain = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), T)  # first call of an unprepared matrix: the CSR kernels
    base = product(plain, api.Descriptor(deterministic=True))
    M = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), T)
    M.set_format(6)  # prepared: the copies are attached before the first product
    runs = [product(M, api.Descriptor(deterministic=True)) for _ in range(4)]
    for got in runs:
        assert np.array_equal(got.view(np.uint32), base.view(np.uint32)), "a deterministic call must not depend on the attached copies or the run"
        assert_same(got, want, "FLOAT", what="deterministic fp32 MULT/PLUS")
    assert_same(product(M, None), want, "FLOAT", what="default fp32 MULT/PLUS on the prepared matrix")
    assert_same(product(M, api.Descriptor()), want, "FLOAT", what="default descriptor after deterministic calls (the setting is per call)")


def test_reference_order_descriptor_gives_the_reference_words(oracle):
    """Descriptor(reference_order=True) (an addition of this build; include/spla.hpp, C: spla_Descriptor_set_reference_order): an fp32 PLUS product through the
    operator API — spla_Exec_mxv_masked and spla_Exec_vxm_masked on a prepared matrix (blocks attached) — is folded as the reference CPU backend folds it
    (src/cpu/cpu_mxv.hpp:94-99, cpu_vxm.hpp:101-113) and equals the oracle BIT FOR BIT; the setting is per call (the next default call is back on the blocks)."""
    api = _api()
    rng = np.random.default_rng(79)
    n_rows, n_cols = 6000, 5000
    Ap, Aj, _ = random_csr(rng, n_rows, n_cols, 14, "FLOAT", skew=True, long_rows=3, long_len=4000)
    nnz = len(Aj)
    Ax = ((rng.random(nnz) - 0.3) * np.exp(rng.normal(0, 3, nnz))).astype(np.float32)
    Ax[Ax == 0] = 1.0
    mask_np = random_mask(rng, n_rows, "FLOAT")
    v_np = ((rng.random(n_cols) - 0.4) * np.exp(rng.normal(0, 2, n_cols))).astype(np.float32)
    v_np[v_np == 0] = 0.5
    T = api.FLOAT
    M = api.Matrix.from_csr(Ap, Aj, Ax, (n_rows, n_cols), T)
    M.set_format(6)
    desc = api.Descriptor(reference_order=True)
    want = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, mask_np, v_np, "MULT", "PLUS", "NQZERO", 0.0)
    mask = api.Vector.from_lists(np.arange(n_rows), mask_np, n_rows, T)
    v = api.Vector.from_lists(np.arange(n_cols), v_np, n_cols, T)
    for _ in range(2):
        got = M.mxv(mask, v, T.MULT, T.PLUS, T.NQZERO, init=api.Scalar(T, 0.0), desc=desc).to_dense()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "reference-order mxv through the operator API differs from the oracle's words"
    # the setting is per call: the next default call re-associates again (these sums cancel, so the bar is relative to the sum of the products' magnitudes)
    dflt = M.mxv(mask, v, T.MULT, T.PLUS, T.NQZERO, init=api.Scalar(T, 0.0)).to_dense()
    mag = oracle.mxv_masked("FLOAT", n_rows, n_cols, Ap, Aj, np.abs(Ax), mask_np, np.abs(v_np), "MULT", "PLUS", "NQZERO", 0.0)
    assert np.all(np.abs(dflt.astype(np.float64) - want) <= 1e-5 * mag + 1e-30), "default call after reference-order calls"
    # vxm: r = u^T M, u dense over the rows, mask over the columns
    u_np = ((rng.random(n_rows) - 0.4) * np.exp(rng.normal(0, 2, n_rows))).astype(np.float32)
    u_np[u_np == 0] = 0.5
    cmask_np = random_mask(rng, n_cols, "FLOAT")
    wi, wx = oracle.vxm_masked("FLOAT", n_rows, n_cols, Ap, Aj, Ax, cmask_np, np.arange(n_rows, dtype=np.uint32), u_np, "MULT", "PLUS", "NQZERO")
    u = api.Vector.from_lists(np.arange(n_rows), u_np, n_rows, T)
    cmask = api.Vector.from_lists(np.arange(n_cols), cmask_np, n_cols, T)
    for mode in (api.Descriptor.PULL, api.Descriptor.PUSH):
        d = api.Descriptor(reference_order=True, traversal_mode=mode)
        ri, rx = u.vxm(cmask, M, T.MULT, T.PLUS, T.NQZERO, init=api.Scalar(T, 0.0), desc=d).read()
        keep = wx != 0  # (entries equal to the fill value 0 are not stored)
        assert np.array_equal(ri, wi[keep]), f"mode {mode}: structure"
        assert np.array_equal(rx.view(np.uint32), wx[keep].view(np.uint32)), f"mode {mode}: reference-order vxm differs from the oracle's words"
