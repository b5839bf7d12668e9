"""Every BASELINE.json config in the -m gpu suite, on the generators SURVEY.md 8d prescribes.

  cfg1  exec_mxv plus-times fp32, uniform random 100k x 100k, 1 M nnz, dense v      (through spla_Exec_mxv_masked, full size)
  cfg2  exec_mxv BAND/BOR int32, R-MAT, mask = BFS depth with EQZERO (the BFS step)   (scale 14 vs oracle: both early_exit values, all plans;
                                                                                       scale 22 full size: row-sampled oracle check + properties)
  cfg3  exec_vxm plus-times fp32, symmetrised R-MAT (stand-in for com-Orkut), 0.85/outdeg values, v = 1/n dense (PageRank step)
  cfg4  exec_mxv min-plus int32, INF = 2^30, weights 1..255, R-MAT; whole and as 2 / 4 / 8 nnz-balanced row blocks stitched
        (scale 15 vs oracle; scale 24 at N = 1 full size: row-sampled oracle check + monotonicity property)
  cfg5  exec_mxm plus-times fp32 A x A and exec_mxmT_masked(L, L, L) int32 (triangle-count form), R-MAT scale 12 vs oracle

Exact checks are written the way the reference writes them (tests/test_mxv.cpp:91-130: every output element compared);
int32 / uint32 bit-exact, fp32 plus-times within 1e-5 relative (north star).
"""
import numpy as np
import pytest
import torch

from helpers import NP

pytestmark = pytest.mark.gpu

FP32_RTOL = 1e-5


def _rmat(scale, ef=16, seed=1, device="cpu", symmetric=False):
    from spla_amd import graphgen
    n = 1 << scale
    key = graphgen.rmat_edge_keys(scale, ef, seed=seed, device=device)
    if symmetric:
        i, j = torch.div(key, n, rounding_mode="floor"), key % n
        key = torch.unique(torch.cat([key, j * n + i]), sorted=True)
    Ap, Aj = graphgen.keys_to_csr(key, n, n)
    return n, Ap, Aj


def _np_u32(t):
    return t.cpu().numpy().view(np.uint32)


def _bfs_depth(n, Ap, Aj, src):
    """depth vector (1 at the source, 0 unreached) of the graph whose edge i -> j is stored at A[i, j]; a pull step r = A x frontier
    discovers row i when some A[i, j] has j in the frontier (reference src/algorithm.cpp:97-99), so levels follow the TRANSPOSED edges."""
    import scipy.sparse as sp
    A = sp.csr_matrix((np.ones(len(Aj), np.int8), Aj.astype(np.int64), Ap.astype(np.int64)), shape=(n, n))
    depth = np.zeros(n, np.int32)
    depth[src] = 1
    front = np.zeros(n, np.int8)
    front[src] = 1
    level = 1
    sizes = []
    while True:
        hit = (A @ front > 0) & (depth == 0)
        if not hit.any():
            break
        level += 1
        depth[hit] = level
        front = hit.astype(np.int8)
        sizes.append(int(hit.sum()))
    return depth, sizes


# ------------------------------------------------------------------------------------------------------------------ cfg1
def test_cfg1_uniform_100k_fp32_plus_times_through_exec_api(oracle):
    """configs[0] at full size through the operator API: spla_Exec_mxv_masked on Matrix / Vector objects (host COO -> AccCsr upload ->
    HIP kernel -> read back), MULT_FLOAT / PLUS_FLOAT, init 0, all rows selected (NQZERO on a mask of ones)."""
    from spla_amd import api, graphgen
    n, nnz = 100_000, 1_000_000
    Ap, Aj = graphgen.uniform_csr(n, nnz, seed=1, device="cpu")
    Ap, Aj = _np_u32(Ap), _np_u32(Aj)
    rng = np.random.default_rng(1)
    Ax = rng.random(nnz, dtype=np.float32)
    v = rng.random(n, dtype=np.float32)
    api.set_accelerator(api.ACCELERATOR_HIP)
    M = api.Matrix.from_csr(Ap, Aj, Ax, (n, n), api.FLOAT)
    hv = api.Vector.from_lists(np.arange(n, dtype=np.uint32), v, n, api.FLOAT)
    hm = api.Vector.from_lists(np.arange(n, dtype=np.uint32), np.ones(n, np.float32), n, api.FLOAT)
    r = M.mxv(hm, hv, api.FLOAT.MULT, api.FLOAT.PLUS, api.FLOAT.NQZERO, init=api.Scalar(api.FLOAT, 0.0))
    ri, rx = r.read()
    got = np.zeros(n, np.float32)
    got[ri] = rx
    want = oracle.mxv_masked("FLOAT", n, n, Ap, Aj, Ax, np.ones(n, np.float32), v, "MULT", "PLUS", "NQZERO", 0.0)
    np.testing.assert_allclose(got, want, rtol=FP32_RTOL, atol=0)
    # rows without an entry keep init = 0 = the fill value: they must be absent from the sparse read-back, like the reference's
    assert set(ri.tolist()) == set(np.nonzero(want != 0)[0].tolist())


# ------------------------------------------------------------------------------------------------------------------ cfg2
@pytest.mark.parametrize("ops", [("BAND", "BOR"), ("LAND", "LOR")])
def test_cfg2_bfs_step_rmat_all_plans_both_early_exit(oracle, ops):
    """configs[1] at an oracle-affordable scale: R-MAT scale 14 (ef 16, directed, loops + duplicates removed), int32 ones, mask = BFS depth
    at the level producing the largest frontier with EQZERO, v = that frontier as dense 0/1, init 0; early_exit false and true; the CSR
    kernels, the sliced-ELL copy and its forced bit form (gpu_utils.hip_mxv cross-checks the three)."""
    from gpu_utils import hip_mxv
    n, Ap, Aj = _rmat(14)
    Ap, Aj = _np_u32(Ap), _np_u32(Aj)
    deg = np.diff(Ap.astype(np.int64))
    depth, sizes = _bfs_depth(n, Ap, Aj, int(np.argmax(deg)))
    L = 1 + int(np.argmax(sizes))
    mask = np.where(depth <= L, depth, 0).astype(np.int32)
    v = (depth == L).astype(np.int32)
    Ax = np.ones(len(Aj), np.int32)
    for early in (False, True):
        want = oracle.mxv_masked("INT", n, n, Ap, Aj, Ax, mask, v, ops[0], ops[1], "EQZERO", 0, early)
        got = hip_mxv("INT", n, n, Ap, Aj, Ax, mask, v, ops[0], ops[1], "EQZERO", 0, early, junk=-7)
        np.testing.assert_array_equal(got, want, err_msg=f"{ops} early_exit={early}")
        assert int((want != 0).sum()) == sizes[L - 1]  # the step discovers exactly the next BFS level


PARITY_REPORT = []  # (test, rows compared, fraction within the plain 1e-5 bar, worst relative difference): printed in the terminal summary (conftest.py)


def _assert_fp32_sum_parity(got, want, exact, min_plain=0.999, what=""):
    """fp32 PLUS folds at full size: the oracle is the reference's sequential fp32 fold, which on a hub row of 10^5 entries is itself ~1e-4 away from the
    sum it approximates, so no reordered fold can agree with it more closely there.  got: HIP, want: oracle, exact: an fp64 fold of the same products
    (float64 arrays).  Every row: |got - want| <= 1e-5 |want| + the oracle's own distance from the fp64 sum; and the plain 1e-5 bar holds on (at least)
    `min_plain` of the rows — all but the longest."""
    got, want, exact = got.astype(np.float64), want.astype(np.float64), exact.astype(np.float64)
    diff = np.abs(got - want)
    assert np.all(diff <= FP32_RTOL * np.abs(want) + 1.01 * np.abs(want - exact) + 1e-30)
    plain = diff <= FP32_RTOL * np.abs(want)
    worst = float(np.max(diff / np.maximum(np.abs(want), 1e-30))) if len(diff) else 0.0
    PARITY_REPORT.append((what, int(len(diff)), float(plain.mean()), worst))
    print(f"[fp32 parity] {what}: {len(diff)} rows against the reference's sequential fold; within plain 1e-5: {plain.mean():.6f}; worst relative difference {worst:.3e}")
    assert plain.mean() >= min_plain, plain.mean()


def _fp64_fold(Ap, Aj, Ax, v):
    """fp64 sum of every row's fp32 products' exact values, on the device (float64 tensor)"""
    n = Ap.numel() - 1
    exact = torch.zeros(n, dtype=torch.float64, device=Ap.device)
    rows = torch.repeat_interleave(torch.arange(n, device=Ap.device), (Ap[1:] - Ap[:-1]).to(torch.int64))
    exact.index_add_(0, rows, Ax.to(torch.float64) * v[Aj.to(torch.int64)].to(torch.float64))
    return exact


def _sampled_rows_oracle(oracle, t, Ap_d, Aj_d, Ax_d, rows_idx, n_cols, mask_d, v_d, mul, add, sel, init, early=False):
    """oracle result of the rows `rows_idx` (device int64 tensor) of a device-resident CSR: rows are independent, so a row sample of the
    full-size result can be checked at full size in seconds."""
    Ap64 = Ap_d.to(torch.int64)
    deg = (Ap64[1:] - Ap64[:-1])[rows_idx]
    Ap_s = torch.zeros(rows_idx.numel() + 1, dtype=torch.int64, device=Ap_d.device)
    torch.cumsum(deg, 0, out=Ap_s[1:])
    nnz_s = int(Ap_s[-1])
    rep = torch.repeat_interleave(torch.arange(rows_idx.numel(), device=Ap_d.device), deg)
    src = Ap64[rows_idx][rep] + (torch.arange(nnz_s, device=Ap_d.device) - Ap_s[:-1][rep])
    Aj_s, Ax_s = Aj_d[src].cpu().numpy().view(np.uint32), Ax_d[src].cpu().numpy()
    return oracle.mxv_masked(t, int(rows_idx.numel()), n_cols, Ap_s.to(torch.int32).cpu().numpy().view(np.uint32), Aj_s, Ax_s,
                             mask_d[rows_idx].cpu().numpy(), v_d.cpu().numpy(), mul, add, sel, init, early)


def test_cfg2_full_size_scale22_properties_and_row_sample(oracle):
    """configs[1] at BASELINE's full size (R-MAT scale 22, ~65 M nnz) through the kernel-level C ABI with the sliced-ELL copy attached (what
    bench.py times): every 64th row against the oracle, plus size-independent properties: unselected rows == init, the result is 0/1,
    early_exit=true gives the same vector (BOR of 0/1 is witness-independent), and the call is idempotent."""
    from spla_amd import kernels
    n, Ap, Aj = _rmat(22, device="cuda")
    nnz = int(Ap[-1])
    ones = torch.ones(nnz, dtype=torch.int32, device="cuda")
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    src = int(deg.argmax())
    plan_csr = kernels.CsrPlan(n, nnz, Ap)
    depth = torch.zeros(n, dtype=torch.int32, device="cuda")
    depth[src] = 1
    front = torch.zeros(n, dtype=torch.int32, device="cuda")
    front[src] = 1
    new = torch.empty(n, dtype=torch.int32, device="cuda")
    sizes, level = [], 1
    while True:
        kernels.mxv_masked(plan_csr, "INT", n, Ap, Aj, ones, depth, front, new, "BAND", "BOR", "EQZERO", 0)
        hit = new != 0
        c = int(hit.sum())
        if c == 0:
            break
        level += 1
        depth[hit] = level
        front = hit.to(torch.int32)
        sizes.append(c)
    L = 1 + int(np.argmax(sizes))
    mask = torch.where(depth <= L, depth, torch.zeros_like(depth)).contiguous()
    v = (depth == L).to(torch.int32)
    plan = kernels.CsrPlan(n, nnz, Ap, n, Aj, ones)
    r = torch.full((n,), -5, dtype=torch.int32, device="cuda")
    kernels.mxv_masked(plan, "INT", n, Ap, Aj, ones, mask, v, r, "BAND", "BOR", "EQZERO", 0)
    torch.cuda.synchronize()
    assert int(((r != 0) & (r != 1)).sum()) == 0
    assert int((r[mask != 0] != 0).sum()) == 0                      # unselected rows hold init
    assert int((r != 0).sum()) == sizes[L - 1]                      # exactly the next level
    r2 = torch.empty_like(r)
    kernels.mxv_masked(plan, "INT", n, Ap, Aj, ones, mask, v, r2, "BAND", "BOR", "EQZERO", 0)
    assert torch.equal(r, r2)                                       # idempotent (double-buffered bitmaps, plan-owned workspaces)
    r3 = torch.empty_like(r)
    kernels.mxv_masked(plan, "INT", n, Ap, Aj, ones, mask, v, r3, "BAND", "BOR", "EQZERO", 0, early_exit=True)
    assert torch.equal(r, r3)
    kernels.mxv_masked(plan_csr, "INT", n, Ap, Aj, ones, mask, v, r3, "BAND", "BOR", "EQZERO", 0)
    assert torch.equal(r, r3)                                       # CSR wave-tile kernel == sliced-ELL bit form
    rows_idx = torch.arange(0, n, 64, device="cuda")
    want = _sampled_rows_oracle(oracle, "INT", Ap, Aj, ones, rows_idx, n, mask, v, "BAND", "BOR", "EQZERO", 0)
    np.testing.assert_array_equal(r[rows_idx].cpu().numpy(), want)
    plan.close()
    plan_csr.close()


# ------------------------------------------------------------------------------------------------------------------ cfg3
def test_cfg3_pagerank_step_vxm_fp32_symmetrised_rmat(oracle):
    """configs[2] on the stated stand-in (com-Orkut is not on the box): symmetrised R-MAT scale 13 ef 28, values 0.85 / outdeg(row)
    (reference examples/pr.cpp:81-88), v = 1/n with every index stored, MULT_FLOAT / PLUS_FLOAT, ALWAYS: the pull form over the CSR of
    M^T, over its sliced-ELL copy and over its column-ordered-blocks copy, and the push form (it keeps the reference's fold order for runs of <= 64 products; longer runs are
    folded by an in-order tree, hence the same 1e-5 bar)."""
    from gpu_utils import dev, host
    from spla_amd import kernels
    n, Ap, Aj = _rmat(13, ef=28, symmetric=True)
    Ap, Aj = _np_u32(Ap), _np_u32(Aj)
    nnz = len(Aj)
    deg = np.diff(Ap.astype(np.int64))
    Ax = np.repeat((0.85 / np.maximum(deg, 1)).astype(np.float32), deg)
    vi, vx = np.arange(n, dtype=np.uint32), np.full(n, 1.0 / n, np.float32)
    mask = np.zeros(n, np.float32)
    wi, wx = oracle.vxm_masked("FLOAT", n, n, Ap, Aj, Ax, mask, vi, vx, "MULT", "PLUS", "ALWAYS")
    dAp, dAj, dAx, dm, dvi, dvx = dev(Ap), dev(Aj), dev(Ax), dev(mask), dev(vi), dev(vx)
    Tp, Tj, Tx = kernels.csr_transpose(n, n, dAp, dAj, dAx)
    for sell in (False, True, "cob"):
        plan_T = kernels.CsrPlan(n, nnz, Tp, n, Tj, Tx) if sell is True else kernels.CsrPlan(n, nnz, Tp)
        if sell == "cob":
            assert plan_T.attach_cob(n, Tp, Tj, Tx) is not None
        ri, rx = kernels.vxm_pull(plan_T, "FLOAT", n, n, Tp, Tj, Tx, dm, dvi, dvx, "MULT", "PLUS", "ALWAYS", identity=True)
        np.testing.assert_array_equal(host(ri, np.uint32), wi, err_msg=f"pull sell={sell}: structure")
        np.testing.assert_allclose(host(rx, np.float32), wx, rtol=FP32_RTOL, atol=0, err_msg=f"pull sell={sell}")
        plan_T.close()
    ri, rx = kernels.vxm_push("FLOAT", n, n, dAp, dAj, dAx, dm, dvi, dvx, "MULT", "PLUS", "ALWAYS")
    np.testing.assert_array_equal(host(ri, np.uint32), wi)
    np.testing.assert_allclose(host(rx, np.float32), wx, rtol=FP32_RTOL, atol=0, err_msg="push")


# ------------------------------------------------------------------------------------------------------------------ cfg4
def _sssp_state(n, rng, finite_frac=0.125):
    """distance vector of a mid-SSSP iteration: INF = 2^30 except a random 1/8 of the vertices with a finite distance"""
    INF = 1 << 30
    d = np.full(n, INF, np.int32)
    k = int(n * finite_frac)
    d[rng.permutation(n)[:k]] = rng.integers(0, 5000, k).astype(np.int32)
    return d, INF


@pytest.mark.parametrize("parts", [1, 2, 4, 8])
def test_cfg4_sssp_step_min_plus_int32_row_blocks_stitched(oracle, parts):
    """configs[3] at scale 15: PLUS_INT (mult) / MIN_INT (add), init INF = 2^30, weights uniform 1..255, ALWAYS; the whole matrix and its
    2 / 4 / 8 nnz-balanced 1-D row blocks (global column ids, the full gathered v — what a rank computes after the all-gather), stitched:
    a row-block split changes no row's fold, so the stitched vector equals the unsharded oracle result bit for bit."""
    from gpu_utils import hip_mxv
    from spla_amd import graphgen
    n, Ap_t, Aj_t = _rmat(15)
    Ap, Aj = _np_u32(Ap_t), _np_u32(Aj_t)
    rng = np.random.default_rng(4)
    Ax = rng.integers(1, 256, len(Aj)).astype(np.int32)
    v, INF = _sssp_state(n, rng)
    mask = np.zeros(n, np.int32)
    want = oracle.mxv_masked("INT", n, n, Ap, Aj, Ax, mask, v, "PLUS", "MIN", "ALWAYS", INF)
    bounds = graphgen.nnz_balanced_row_splits(Ap_t, parts)
    got = np.empty(n, np.int32)
    for p in range(parts):
        lo, hi = bounds[p], bounds[p + 1]
        if hi == lo:
            continue
        s, e = int(Ap[lo]), int(Ap[hi])
        Ap_b = (Ap[lo:hi + 1].astype(np.int64) - s).astype(np.uint32)
        got[lo:hi] = hip_mxv("INT", hi - lo, n, Ap_b, Aj[s:e].copy(), Ax[s:e].copy(), mask[lo:hi], v, "PLUS", "MIN", "ALWAYS", INF, junk=3)
    np.testing.assert_array_equal(got, want)


def test_cfg4_full_size_scale24_row_sample_and_monotone(oracle):
    """configs[3] at N = 1, full size (R-MAT scale 24, ~256 M nnz, 2.1 GB CSR): sliced-ELL copy attached; every 512th row against the oracle;
    properties: r <= INF everywhere, r == INF exactly on rows none of whose neighbours is finite, and relaxing a second time with
    v' = min(v, r) can only lower a distance (min-plus is monotone)."""
    from spla_amd import kernels
    n, Ap, Aj = _rmat(24, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(44)
    Ax = torch.randint(1, 256, (nnz,), generator=gen, device="cuda", dtype=torch.int32)
    INF = 1 << 30
    v = torch.full((n,), INF, dtype=torch.int32, device="cuda")
    pos = torch.randint(0, n, (n // 8,), generator=gen, device="cuda")
    v[pos] = torch.randint(0, 5000, (n // 8,), generator=gen, device="cuda", dtype=torch.int32)
    mask = torch.zeros(n, dtype=torch.int32, device="cuda")
    plan = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax)
    r = torch.empty(n, dtype=torch.int32, device="cuda")
    kernels.mxv_masked(plan, "INT", n, Ap, Aj, Ax, mask, v, r, "PLUS", "MIN", "ALWAYS", INF)
    torch.cuda.synchronize()
    assert int((r > INF).sum()) == 0 and int((r < 1).sum()) == 0
    rows_idx = torch.arange(0, n, 512, device="cuda")
    want = _sampled_rows_oracle(oracle, "INT", Ap, Aj, Ax, rows_idx, n, mask, v, "PLUS", "MIN", "ALWAYS", INF)
    np.testing.assert_array_equal(r[rows_idx].cpu().numpy(), want)
    # rows with no finite neighbour stay at INF, all others drop below it: structure-only product over the 0/1 "finite" vector
    fin = (v < INF).to(torch.int32)
    ones = torch.ones(nnz, dtype=torch.int32, device="cuda")
    plan_csr = kernels.CsrPlan(n, nnz, Ap)
    reach = torch.empty(n, dtype=torch.int32, device="cuda")
    kernels.mxv_masked(plan_csr, "INT", n, Ap, Aj, ones, mask, fin, reach, "BAND", "BOR", "ALWAYS", 0)
    assert torch.equal(reach != 0, r < INF)
    v2 = torch.minimum(v, r)
    r2 = torch.empty_like(r)
    kernels.mxv_masked(plan, "INT", n, Ap, Aj, Ax, mask, v2, r2, "PLUS", "MIN", "ALWAYS", INF)
    assert int((r2 > r).sum()) == 0
    plan.close()
    plan_csr.close()


# ------------------------------------------------------------------------------------------------------------------ cfg5
def test_cfg5_mxm_fp32_rmat_AxA(oracle):
    """configs[4], A x A: R-MAT scale 12 fp32 U(0,1) values, MULT / PLUS, init 0 (the oracle's dense accumulator row is O(n_cols) per row:
    scale 12 keeps it in seconds).  Structure identical; values within 1e-5 (runs longer than 64 products are folded as a tree)."""
    from gpu_utils import dev, host
    from spla_amd import kernels
    n, Ap, Aj = _rmat(12)
    Ap, Aj = _np_u32(Ap), _np_u32(Aj)
    rng = np.random.default_rng(5)
    Ax = (rng.random(len(Aj), dtype=np.float32) + np.float32(0.01))
    wp, wj, wx = oracle.mxm("FLOAT", n, n, (Ap, Aj, Ax), (Ap, Aj, Ax), "MULT", "PLUS", 0.0)
    A = (dev(Ap), dev(Aj), dev(Ax))
    for block in (0, 1 << 16):  # default row blocks, and many small ones
        (Rp, Rj, Rx), products = kernels.mxm("FLOAT", n, n, A, A, "MULT", "PLUS", 0.0, block_products=block)
        deg = np.diff(Ap.astype(np.int64))
        assert products == int(deg[Aj].sum())
        np.testing.assert_array_equal(host(Rp, np.uint32), wp)
        np.testing.assert_array_equal(host(Rj, np.uint32), wj)
        np.testing.assert_allclose(host(Rx, np.float32), wx, rtol=FP32_RTOL, atol=0)


def test_cfg5_triangle_count_form_mxmT_LLL(oracle):
    """configs[4], triangle-count path: L = strictly-lower triangle of the symmetrised R-MAT scale 12 pattern, int32 ones,
    exec_mxmT_masked(L, L, L) with MULT / PLUS / GTZERO, init 0 (reference src/algorithm.cpp:398, examples/tc.cpp:82-86); bit-exact,
    and the reduced sum equals the triangle count obtained from the dense A^3 trace."""
    from gpu_utils import dev, host
    from spla_amd import kernels
    import scipy.sparse as sp
    n, Ap, Aj = _rmat(12, symmetric=True)
    Ap, Aj = _np_u32(Ap), _np_u32(Aj)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap.astype(np.int64)))
    keep = Aj.astype(np.int64) < rows
    Lj = Aj[keep]
    Lp = np.zeros(n + 1, np.uint32)
    Lp[1:] = np.cumsum(np.bincount(rows[keep], minlength=n))
    Lx = np.ones(len(Lj), np.int32)
    Lt = (Lp, Lj, Lx)
    wp, wj, wx = oracle.mxmT_masked("INT", n, Lt, Lt, Lt, "MULT", "PLUS", "GTZERO", 0)
    L = (dev(Lp), dev(Lj), dev(Lx))
    Rp, Rj, Rx = kernels.mxmT_masked("INT", n, L, L, L, "MULT", "PLUS", "GTZERO", 0)
    np.testing.assert_array_equal(host(Rp, np.uint32), wp)
    np.testing.assert_array_equal(host(Rj, np.uint32), wj)
    np.testing.assert_array_equal(host(Rx, np.int32), wx)
    Ls = sp.csr_matrix((Lx.astype(np.int64), Lj.astype(np.int64), Lp.astype(np.int64)), shape=(n, n))
    tri = int((Ls @ Ls.T).multiply(Ls).sum())
    assert int(host(Rx, np.int32).astype(np.int64).sum()) == tri


# ------------------------------------------------------------------------------------------ full-size parity of the column-ordered blocks
# (the layout every headline number rests on since round 2: hub + natural ranges, cost-cut pieces, shared accumulators and tickets at the
# geometry the bench runs — not the forced small geometry of the unit tests)
def test_cfg4_full_size_scale24_column_ordered_blocks(oracle):
    """configs[3] at N = 1, full size (R-MAT scale 24, ~256 M nnz) over the column-ordered-blocks copy: the value form and the sparse-input-vector
    form of min-plus (SPLAHIP_COB_BITS=2 forces it; at 2^24 columns the library picks it itself), each bit-identical to the CSR kernel on ALL rows
    and, every 512th row, to the oracle; the fp32 form of the same step (the reference's own SSSP semiring) bit-identical too."""
    import os
    from spla_amd import kernels
    n, Ap, Aj = _rmat(24, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(44)
    Ax = torch.randint(1, 256, (nnz,), generator=gen, device="cuda", dtype=torch.int32)
    INF = 1 << 30
    v = torch.full((n,), INF, dtype=torch.int32, device="cuda")
    pos = torch.randint(0, n, (n // 8,), generator=gen, device="cuda")
    v[pos] = torch.randint(0, 5000, (n // 8,), generator=gen, device="cuda", dtype=torch.int32)
    mask = torch.zeros(n, dtype=torch.int32, device="cuda")
    plan_csr = kernels.CsrPlan(n, nnz, Ap)
    r_csr = torch.empty(n, dtype=torch.int32, device="cuda")
    kernels.mxv_masked(plan_csr, "INT", n, Ap, Aj, Ax, mask, v, r_csr, "PLUS", "MIN", "ALWAYS", INF)
    rows_idx = torch.arange(0, n, 512, device="cuda")
    want = _sampled_rows_oracle(oracle, "INT", Ap, Aj, Ax, rows_idx, n, mask, v, "PLUS", "MIN", "ALWAYS", INF)
    np.testing.assert_array_equal(r_csr[rows_idx].cpu().numpy(), want)
    plan = kernels.CsrPlan(n, nnz, Ap)
    info = plan.attach_cob(n, Ap, Aj, Ax)
    assert info and info["hub_rows"] > 0 and info["ranges_in_pieces"] > 0, info
    assert "mxv_cob" in kernels.mxv_variant("INT", "PLUS", "MIN", False, plan)
    for bits in ("0", "2"):  # value form, sparse-input-vector form
        os.environ["SPLAHIP_COB_BITS"] = bits
        for rep in range(2):  # twice: the shared accumulators and tickets must come back armed
            r = torch.full((n,), -5, dtype=torch.int32, device="cuda")
            kernels.mxv_masked(plan, "INT", n, Ap, Aj, Ax, mask, v, r, "PLUS", "MIN", "ALWAYS", INF)
            torch.cuda.synchronize()
            assert torch.equal(r, r_csr), f"column-ordered blocks (SPLAHIP_COB_BITS={bits}, call {rep}) differ from the CSR kernel"
    os.environ["SPLAHIP_COB_BITS"] = "1"
    plan.close()
    # fp32 min-plus, init = +inf (reference src/algorithm.cpp:208-213)
    Axf = Ax.to(torch.float32)
    vf = torch.where(v == INF, torch.full((n,), float("inf"), device="cuda"), v.to(torch.float32))
    maskf = torch.zeros(n, dtype=torch.float32, device="cuda")
    rf_csr = torch.empty(n, dtype=torch.float32, device="cuda")
    kernels.mxv_masked(plan_csr, "FLOAT", n, Ap, Aj, Axf, maskf, vf, rf_csr, "PLUS", "MIN", "ALWAYS", float("inf"))
    wantf = _sampled_rows_oracle(oracle, "FLOAT", Ap, Aj, Axf, rows_idx, n, maskf, vf, "PLUS", "MIN", "ALWAYS", float("inf"))
    np.testing.assert_array_equal(rf_csr[rows_idx].cpu().numpy().view(np.uint32), wantf.view(np.uint32))
    planf = kernels.CsrPlan(n, nnz, Ap)
    planf.attach_cob(n, Ap, Aj, Axf)
    assert "integer keys" in kernels.mxv_variant("FLOAT", "PLUS", "MIN", False, planf)
    rf = torch.empty_like(rf_csr)
    kernels.mxv_masked(planf, "FLOAT", n, Ap, Aj, Axf, maskf, vf, rf, "PLUS", "MIN", "ALWAYS", float("inf"))
    torch.cuda.synchronize()
    assert torch.equal(rf.view(torch.int32), rf_csr.view(torch.int32))
    # the integer and the float step agree where both are exact (weights and distances are small integers)
    assert torch.equal(torch.where(torch.isinf(rf), torch.full_like(rf, float(INF)), rf).to(torch.int32), r_csr)
    planf.close()
    plan_csr.close()


def test_north_star_full_size_scale23_fp32_column_ordered_blocks(oracle):
    """The north-star configuration at full size: fp32 MULT / PLUS over all rows of R-MAT scale 23 (131 M nnz) over the column-ordered blocks.
    Every row against an fp64 fold of the same products (2e-6 relative: the kernel sums in double and rounds once per row and piece).  Every 64th
    row against the oracle — the reference's sequential fp32 fold — at north_star's 1e-5 relative PLUS the oracle's own distance from the fp64 sum
    on that row: a hub row of 10^5 entries folded one by one in fp32 is itself 1e-4 away from the sum it approximates, and no reordering can agree
    with it more closely than that (rows where the sequential fold is accurate, the great majority, are held to the plain 1e-5).  A masked call
    (NQZERO on a random mask: unselected rows == init exactly, selected rows unchanged) and two calls agreeing to rounding."""
    from spla_amd import kernels
    n, Ap, Aj = _rmat(23, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(401)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    plan = kernels.CsrPlan(n, nnz, Ap)
    info = plan.attach_cob(n, Ap, Aj, Ax)
    assert info and info["hub_rows"] > 0 and info["ranges_in_pieces"] > 0, info
    assert "fp64 LDS accumulation" in kernels.mxv_variant("FLOAT", "MULT", "PLUS", False, plan)
    r = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0)
    torch.cuda.synchronize()
    rows_idx = torch.arange(0, n, 64, device="cuda")
    want = _sampled_rows_oracle(oracle, "FLOAT", Ap, Aj, Ax, rows_idx, n, mask, v, "MULT", "PLUS", "ALWAYS", 0.0)
    exact = _fp64_fold(Ap, Aj, Ax, v)
    rel = ((r.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
    assert rel < 2e-6, rel  # one rounding per row and piece on top of an exact double sum
    _assert_fp32_sum_parity(r[rows_idx].cpu().numpy(), want, exact[rows_idx].cpu().numpy(), what="configs[1] fp32 plus-times, R-MAT 23, every 256th row")
    assert int((r[(Ap[1:] == Ap[:-1])] != 0).sum()) == 0  # rows without an entry: init
    r2 = torch.empty_like(r)
    kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r2, "MULT", "PLUS", "ALWAYS", 0.0)
    assert float(((r2 - r).abs() / r.abs().clamp_min(1e-30)).max()) < 1e-6
    m2 = (torch.rand(n, generator=gen, device="cuda") < 0.5).to(torch.float32)
    r3 = torch.empty_like(r)
    kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, m2, v, r3, "MULT", "PLUS", "NQZERO", 7.5)
    torch.cuda.synchronize()
    assert bool(torch.all(r3[m2 == 0] == 7.5))
    sel = m2 != 0
    assert float((((r3[sel] - 7.5) - r[sel]).abs() / r[sel].abs().clamp_min(1.0)).max()) < 1e-5
    # the bench's masked variants at full size (reference src/cpu/cpu_mxv.hpp:89-92: an unselected row is init and its entries are never read): NQZERO masks
    # selecting 10 % / 50 % of the rows at random and 10 % / 50 % of the rows below hub degree; each call twice (the second runs the side its own probe
    # favours: the selected-rows kernels or the blocks), every 256th row against the oracle, unselected rows == init exactly
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    served = {}
    for name, frac, light in (("10 % random", 0.1, False), ("50 % random", 0.5, False), ("10 % no hub row", 0.1, True), ("50 % no hub row", 0.5, True)):
        pick = torch.rand(n, generator=gen, device="cuda") < frac
        if light:
            pick &= deg < 31
        m = pick.to(torch.float32)
        rm = torch.empty_like(r)
        for _ in range(2):
            rm.fill_(-2.0)
            kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, m, v, rm, "MULT", "PLUS", "NQZERO", 0.25)
            torch.cuda.synchronize()
        served[name] = plan.mask_probe()["served_by_rows"]
        assert bool(torch.all(rm[~pick] == 0.25)), name
        rows_m = torch.arange(0, n, 256, device="cuda")
        want_m = _sampled_rows_oracle(oracle, "FLOAT", Ap, Aj, Ax, rows_m, n, m, v, "MULT", "PLUS", "NQZERO", 0.25)
        exact_m = torch.where(pick, exact + 0.25, torch.full_like(exact, 0.25))
        _assert_fp32_sum_parity(rm[rows_m].cpu().numpy(), want_m, exact_m[rows_m].cpu().numpy(), what=f"configs[1] masked, {name} (served by {'rows' if served[name] else 'blocks'})")
    assert served["10 % no hub row"] and not served["50 % random"], served
    plan.close()


def test_cfg3_full_size_pagerank_pull_column_ordered_blocks(oracle):
    """configs[2] at the stated stand-in's full size (symmetrised R-MAT scale 22, edge factor 28: ~220 M nnz; com-Orkut is not on the box): the pull
    form of exec_vxm_masked over the column-ordered blocks of M^T; every output index present (the graph is symmetric: a column is touched iff its
    row holds an entry), values of every 256th output against the oracle's mxv over M^T at 1e-5 (vxm r = v^T M is mxv over M^T with mult(v, x))."""
    from spla_amd import kernels
    n, Ap, Aj = _rmat(22, ef=28, device="cuda", symmetric=True)
    nnz = int(Ap[-1])
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    Ax = (0.85 / deg.clamp(min=1).to(torch.float32))[rows].contiguous()
    del rows
    Tp, Tj, Tx = kernels.csr_transpose(n, n, Ap, Aj, Ax)
    plan_T = kernels.CsrPlan(n, nnz, Tp)
    assert plan_T.attach_cob(n, Tp, Tj, Tx) is not None
    vi = torch.arange(n, dtype=torch.int32, device="cuda")
    vx = torch.full((n,), 1.0 / n, dtype=torch.float32, device="cuda")
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    ri, rx = kernels.vxm_pull(plan_T, "FLOAT", n, n, Tp, Tj, Tx, mask, vi, vx, "MULT", "PLUS", "ALWAYS", identity=True)
    torch.cuda.synchronize()
    touched = torch.nonzero(Tp[1:] != Tp[:-1]).flatten().to(torch.int32)
    assert torch.equal(ri.to(torch.int32), touched)
    dense = torch.zeros(n, dtype=torch.float32, device="cuda")
    dense[ri.to(torch.int64)] = rx
    rows_idx = torch.arange(0, n, 256, device="cuda")
    want = _sampled_rows_oracle(oracle, "FLOAT", Tp, Tj, Tx, rows_idx, n, mask, vx, "MULT", "PLUS", "ALWAYS", 0.0)  # x * v == v * x in fp32
    exact = _fp64_fold(Tp, Tj, Tx, vx)
    assert float(((dense.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max()) < 2e-6
    # (PageRank products are tiny and of mixed magnitude: the reference's sequential fold strays from the sum on more rows than with U(0, 1) values —
    # measured on this input: 99.35 % of the sampled outputs within plain 1e-5, the worst 1.8e-4 away, all of them inside the reference's own distance
    # from the fp64 sum)
    _assert_fp32_sum_parity(dense[rows_idx].cpu().numpy(), want, exact[rows_idx].cpu().numpy(), min_plain=0.98, what="configs[2] PageRank pull step, every 256th output")
    plan_T.close()


def _rows_sample_with_the_heaviest(Ap, step, n_heavy):
    """every `step`-th row plus the `n_heavy` longest rows (the rows on which a re-associated fp32 fold strays furthest from the reference's)"""
    n = Ap.numel() - 1
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    heavy = torch.topk(deg, n_heavy).indices
    return torch.unique(torch.cat([torch.arange(0, n, step, device=Ap.device), heavy]))


def test_north_star_full_size_reference_order_is_bit_identical_to_the_oracle(oracle):
    """the north-star call — fp32 MULT / PLUS over all rows of R-MAT scale 23 (131 M nnz) — in REFERENCE-ORDER mode
    (splahip_set_reference_order; Descriptor::set_reference_order): every row is folded strictly left to right from init in entry order, the loop of the
    reference CPU backend (src/cpu/cpu_mxv.hpp:94-99), so there is no tolerance here: every 64th row and the 4096 longest rows (up to ~10^5 entries each) must
    carry the oracle's WORD.  The plan holds the blocks (they stand aside for the call).  Masked once more (NQZERO, init 7.5).  The call's time is printed."""
    import time
    from spla_amd import _lib, kernels
    n, Ap, Aj = _rmat(23, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(401)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    mask = (torch.rand(n, generator=gen, device="cuda") < 0.5).to(torch.float32)
    plan = kernels.CsrPlan(n, nnz, Ap)
    assert plan.attach_cob(n, Ap, Aj, Ax)
    rows_idx = _rows_sample_with_the_heaviest(Ap, 64, 4096)
    r = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    assert _lib.call("splahip_set_reference_order", 1) == 0
    try:
        assert "mxv_ordered" in kernels.mxv_variant("FLOAT", "MULT", "PLUS", False, plan)
        for (sel, init) in (("ALWAYS", 0.0), ("NQZERO", 7.5)):
            kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", sel, init)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", sel, init)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            want = _sampled_rows_oracle(oracle, "FLOAT", Ap, Aj, Ax, rows_idx, n, mask, v, "MULT", "PLUS", sel, init)
            got = r[rows_idx].cpu().numpy()
            bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
            print(f"[reference order] configs[1] R-MAT 23 fp32 plus-times {sel}: {len(got)} rows (the 4096 longest included) bit-identical to the oracle: {len(bad) == 0}; call {ms:.2f} ms")
            PARITY_REPORT.append((f"configs[1] R-MAT 23, reference-order mode, {sel}: bit-identical rows", int(len(got)), float((len(got) - len(bad)) / len(got)), 0.0 if len(bad) == 0 else float("nan")))
            assert len(bad) == 0, f"{sel}: {len(bad)} of {len(got)} rows differ from the oracle's word; first: row {int(rows_idx[bad[0]])}: {got[bad[0]]!r} vs {want[bad[0]]!r}"
    finally:
        assert _lib.call("splahip_set_reference_order", 0) == 1
    plan.close()


def test_cfg3_full_size_pagerank_pull_reference_order_is_bit_identical_to_the_oracle(oracle):
    """VERDICT r5 item 2, configs[2]: the PageRank step — the pull form of exec_vxm_masked over M^T of the symmetrised R-MAT scale 22 stand-in (~220 M nnz),
    values 0.85 / outdeg, v = 1/n — in reference-order mode: every output index adds its products in the order of v's indices (src/cpu/cpu_vxm.hpp:101-113).
    Every 256th output and the 2048 longest columns against the oracle, bit for bit (the default mode holds 99.35 % of these outputs inside plain 1e-5)."""
    import time
    from spla_amd import _lib, kernels
    n, Ap, Aj = _rmat(22, ef=28, device="cuda", symmetric=True)
    nnz = int(Ap[-1])
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    Ax = (0.85 / deg.clamp(min=1).to(torch.float32))[rows].contiguous()
    del rows
    Tp, Tj, Tx = kernels.csr_transpose(n, n, Ap, Aj, Ax)
    plan_T = kernels.CsrPlan(n, nnz, Tp)
    assert plan_T.attach_cob(n, Tp, Tj, Tx) is not None
    vi = torch.arange(n, dtype=torch.int32, device="cuda")
    vx = torch.full((n,), 1.0 / n, dtype=torch.float32, device="cuda")
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    assert _lib.call("splahip_set_reference_order", 1) == 0
    try:
        ri, rx = kernels.vxm_pull(plan_T, "FLOAT", n, n, Tp, Tj, Tx, mask, vi, vx, "MULT", "PLUS", "ALWAYS", identity=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ri, rx = kernels.vxm_pull(plan_T, "FLOAT", n, n, Tp, Tj, Tx, mask, vi, vx, "MULT", "PLUS", "ALWAYS", identity=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    finally:
        assert _lib.call("splahip_set_reference_order", 0) == 1
    touched = torch.nonzero(Tp[1:] != Tp[:-1]).flatten().to(torch.int32)
    assert torch.equal(ri.to(torch.int32), touched)
    dense = torch.zeros(n, dtype=torch.float32, device="cuda")
    dense[ri.to(torch.int64)] = rx
    rows_idx = _rows_sample_with_the_heaviest(Tp, 256, 2048)
    want = _sampled_rows_oracle(oracle, "FLOAT", Tp, Tj, Tx, rows_idx, n, mask, vx, "MULT", "PLUS", "ALWAYS", 0.0)  # x * v == v * x; 0 + p == p for p > 0
    got = dense[rows_idx].cpu().numpy()
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    print(f"[reference order] configs[2] PageRank pull step: {len(got)} outputs (the 2048 longest columns included) bit-identical to the oracle: {len(bad) == 0}; call {ms:.2f} ms")
    PARITY_REPORT.append(("configs[2] PageRank pull step, reference-order mode: bit-identical outputs", int(len(got)), float((len(got) - len(bad)) / len(got)), 0.0 if len(bad) == 0 else float("nan")))
    assert len(bad) == 0, f"{len(bad)} of {len(got)} outputs differ from the oracle's word; first: {got[bad[0]]!r} vs {want[bad[0]]!r}"
    plan_T.close()


# ------------------------------------------------------------------------------------------ configs[4] and the N = 8 split of configs[3] at BASELINE's full sizes
def test_cfg5_full_size_mxmT_rmat20_row_sample_and_triangle_count(oracle):
    """configs[4], triangle-count form at full size: L = strict lower triangle of symmetrised R-MAT scale 20 (15.7 M entries), exec_mxmT_masked(L, L, L) MULT / PLUS /
    GTZERO int32.  The triangle count is the known answer of this graph (423 516 054: the reference CPU backend's and the oracle's tc on it, profiles/r0*_ops_algorithms.json);
    the same call with the mask restricted to every 64th row is checked entry for entry against the oracle (rows are independent), and the full result's rows agree with it."""
    from spla_amd import graphgen, kernels
    n, Ap, Aj = _rmat(20, device="cuda", symmetric=True)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
    low = Aj.to(torch.int64) < rows
    Lp, Lj = graphgen.keys_to_csr(rows[low] * n + Aj[low].to(torch.int64), n, n)
    del rows, low
    Lx = torch.ones(int(Lp[-1]), dtype=torch.int32, device="cuda")
    L = (Lp, Lj, Lx)
    Rp, Rj, Rx = kernels.mxmT_masked("INT", n, L, L, L, "MULT", "PLUS", "GTZERO", 0)
    assert int(Rx.to(torch.int64).sum()) == 423516054
    keep = torch.zeros(n, dtype=torch.bool, device="cuda")
    keep[::64] = True
    Lp64 = Lp.to(torch.int64)
    deg = Lp64[1:] - Lp64[:-1]
    Mp = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(torch.where(keep, deg, torch.zeros_like(deg)), 0, out=Mp[1:])
    ent = torch.repeat_interleave(keep, deg)
    M = (Mp.to(torch.int32), Lj[ent].contiguous(), Lx[ent].contiguous())
    Sp, Sj, Sx = kernels.mxmT_masked("INT", n, M, L, L, "MULT", "PLUS", "GTZERO", 0)
    Lh = tuple(_np_u32(t) if t.dtype == torch.int32 and i < 2 else t.cpu().numpy() for i, t in enumerate(L))
    Mh = (_np_u32(M[0]), _np_u32(M[1]), M[2].cpu().numpy())
    wp, wj, wx = oracle.mxmT_masked("INT", n, Mh, Lh, Lh, "MULT", "PLUS", "GTZERO", 0)
    np.testing.assert_array_equal(_np_u32(Sp), wp)
    np.testing.assert_array_equal(_np_u32(Sj), wj)
    np.testing.assert_array_equal(Sx.cpu().numpy(), wx)
    # the sampled rows of the full result are the same rows
    Rp64, Sp64 = Rp.to(torch.int64), Sp.to(torch.int64)
    sel = torch.nonzero(keep).flatten()
    assert torch.equal((Rp64[1:] - Rp64[:-1])[sel], (Sp64[1:] - Sp64[:-1])[sel])
    full_ent = torch.repeat_interleave(keep, Rp64[1:] - Rp64[:-1])
    assert torch.equal(Rj[full_ent], Sj) and torch.equal(Rx[full_ent], Sx)


@pytest.mark.parametrize("scale", [18, 19])
def test_cfg5_full_size_mxm_rmat_row_sample(oracle, scale):
    """configs[4], A x A, fp32 MULT / PLUS, at the bench's two sizes: R-MAT scale 18 (3.9 M entries, 2.9 G products, 1.28 G result entries) and scale 19 — 3.53 G result
    entries, the LARGEST result of this family the reference's 32-bit row pointers (include/spla/config.hpp:56) can index (scale 20 of BASELINE configs[4] exceeds 2^32: the
    call refuses it, tests/test_mxm_gpu.py).  The product count, the row pointers' total, and every 1024th row of A against the oracle (a row of the result depends on that
    row of A and on B only): structure identical, values within 1e-5."""
    from spla_amd import kernels
    n, Ap, Aj = _rmat(scale, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1809)
    Ax = (torch.rand(nnz, generator=gen, device="cuda") * 0.5 + 0.25).contiguous()
    A = (Ap, Aj, Ax)
    raw, products = kernels.mxm("FLOAT", n, n, A, A, "MULT", "PLUS", 0.0, adopt=False)
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    assert products == int(deg[Aj.to(torch.int64)].sum())
    Rp = raw.Rp.to(torch.int64) & 0xFFFFFFFF  # (row pointers are 32-bit unsigned words: past 2^31 at scale 19)
    assert int(Rp[-1]) == raw.nnz and raw.nnz > (1_000_000_000 if scale == 18 else 3_000_000_000)
    assert bool(torch.all(Rp[1:] >= Rp[:-1]))
    rows_idx = torch.arange(0, n, 1024, device="cuda")
    Ap64 = Ap.to(torch.int64)
    d = deg[rows_idx]
    Sp = torch.zeros(rows_idx.numel() + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(d, 0, out=Sp[1:])
    rep = torch.repeat_interleave(torch.arange(rows_idx.numel(), device="cuda"), d)
    src = Ap64[rows_idx][rep] + (torch.arange(int(Sp[-1]), device="cuda") - Sp[:-1][rep])
    sub = (_np_u32(Sp.to(torch.int32)), _np_u32(Aj[src]), Ax[src].cpu().numpy())
    wp, wj, wx = oracle.mxm("FLOAT", int(rows_idx.numel()), n, sub, (_np_u32(Ap), _np_u32(Aj), Ax.cpu().numpy()), "MULT", "PLUS", 0.0)
    Rj, Rx = raw.adopt("FLOAT")[1:]
    lens = (Rp[1:] - Rp[:-1])[rows_idx]
    np.testing.assert_array_equal(lens.cpu().numpy(), np.diff(wp.astype(np.int64)))
    # the sampled rows' entries by explicit positions (the result has more than 2^31 entries at scale 19: no boolean mask over it)
    Op = torch.zeros(rows_idx.numel() + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(lens, 0, out=Op[1:])
    orep = torch.repeat_interleave(torch.arange(rows_idx.numel(), device="cuda"), lens)
    pos = Rp[rows_idx][orep] + (torch.arange(int(Op[-1]), device="cuda") - Op[:-1][orep])
    np.testing.assert_array_equal(_np_u32(Rj[pos]), wj)
    np.testing.assert_allclose(Rx[pos].cpu().numpy(), wx, rtol=FP32_RTOL, atol=0)
    del Rj, Rx, pos, raw
    torch.cuda.empty_cache()
    from spla_amd import _lib
    _lib.call("splahip_mxm_trim")


def test_cfg4_full_size_scale24_eight_way_split_stitched(oracle):
    """configs[3] as BASELINE states it — R-MAT scale 24 across 8 ranks — with the eight nnz-balanced row blocks run one after the other on this one GPU through the
    C-ABI shard (the split into local-column and remote-column parts with their own column-ordered blocks, splahip_shard_mxv_local / _remote over the gathered vector):
    the stitched result equals the unsplit product over the blocks on every row, and the oracle on every 1024th row.  Owner-sorted labels, as `--gpus 8` runs."""
    from spla_amd import graphgen, kernels, sharding
    n, Ap, Aj = graphgen.rmat_csr(24, 16, seed=1, device="cuda", permute="owner-sorted", owners=8)
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2408)
    Ax = torch.randint(1, 256, (nnz,), generator=gen, device="cuda", dtype=torch.int32)
    INF = 1 << 30
    v = torch.full((n,), INF, dtype=torch.int32, device="cuda")
    v[torch.randint(0, n, (n // 8,), generator=gen, device="cuda")] = torch.randint(0, 5000, (n // 8,), generator=gen, device="cuda", dtype=torch.int32)
    mask = torch.zeros(n, dtype=torch.int32, device="cuda")
    plan = kernels.CsrPlan(n, nnz, Ap)
    assert plan.attach_cob(n, Ap, Aj, Ax) is not None
    whole = torch.empty(n, dtype=torch.int32, device="cuda")
    kernels.mxv_masked(plan, "INT", n, Ap, Aj, Ax, mask, v, whole, "PLUS", "MIN", "ALWAYS", INF)
    torch.cuda.synchronize()
    plan.close()
    rows_idx = torch.arange(0, n, 1024, device="cuda")
    want = _sampled_rows_oracle(oracle, "INT", Ap, Aj, Ax, rows_idx, n, mask, v, "PLUS", "MIN", "ALWAYS", INF)
    np.testing.assert_array_equal(whole[rows_idx].cpu().numpy(), want)
    world = 8
    bounds = sharding.row_splits(Ap, world)
    got = torch.empty(n, dtype=torch.int32, device="cuda")
    for rank in range(world):
        blk = sharding.RowBlock(Ap, Aj, Ax, bounds, rank)
        sh = sharding.Shard(blk, blk.Ax)
        v_full = torch.full((blk.n_cols_padded,), INF, dtype=torch.int32, device="cuda")
        for p in range(world):  # what the all-gather of the padded slices leaves on every rank
            v_full[p * blk.max_rows: p * blk.max_rows + (bounds[p + 1] - bounds[p])] = v[bounds[p]:bounds[p + 1]]
        v_loc = blk.pad_local(v[blk.lo:blk.hi], fill=INF)
        r = torch.empty(blk.n_local, dtype=torch.int32, device="cuda")
        m = mask[blk.lo:blk.hi].clone()  # (a fresh allocation: the remote part moves mask and r in 16-byte loads)
        if sh.serves("INT", "MIN"):
            sh.mxv_local("INT", m, v_loc, r, "PLUS", "MIN", "ALWAYS", INF)
            sh.mxv_remote("INT", m, v_full, r, "PLUS", "MIN", "ALWAYS")
        else:
            sh.mxv_whole("INT", m, v_full, r, "PLUS", "MIN", "ALWAYS", INF)
        torch.cuda.synchronize()
        got[blk.lo:blk.hi] = r
        sh.close()
        del blk, sh, v_full
    assert torch.equal(got, whole)
