"""The 1-D row-block decomposition behind the kernel-level C ABI (spla_amd/csrc/kernels/dist.hip) on ONE GPU: the shards of every rank are built
and driven one after the other — local-column part, "exchange" (a device-side concatenation here), remote-column part — and the stitched result
must equal the unsharded oracle result: a row-block split changes no row's fold (bit-exact for int32; 1e-5 for fp32 plus-times).  The RCCL
communicator is exercised at world size 1 (a single-GPU box cannot host two RCCL ranks); the N > 1 exchange itself is covered by the gloo tests on
CPU and by bench.py --gpus N on a multi-GPU node."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(scale=13, ef=16):
    from spla_amd import graphgen
    n, Ap, Aj = graphgen.rmat_csr(scale, ef, seed=3, device="cpu")
    return n, Ap, Aj


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_phases_equal_whole_min_plus_int32(oracle, world):
    """BASELINE configs[3] in its sharded form: min-plus int32, INF = 2^30, weights 1..255, 2 / 3 / 8 nnz-balanced row blocks."""
    from spla_amd import sharding
    n, Ap, Aj = _graph()
    g = torch.Generator().manual_seed(11)
    w = torch.randint(1, 256, (int(Ap[-1]),), generator=g, dtype=torch.int32)
    INF = 1 << 30
    d = torch.full((n,), INF, dtype=torch.int32)
    d[torch.randperm(n, generator=g)[: n // 8]] = torch.randint(0, 5000, (n // 8,), generator=g, dtype=torch.int32)
    mask = torch.zeros(n, dtype=torch.int32)
    want = oracle.mxv_masked("INT", n, n, Ap.numpy(), Aj.numpy(), w.numpy(), mask.numpy(), d.numpy(), "PLUS", "MIN", "ALWAYS", INF)
    bounds = sharding.row_splits(Ap, world)
    blks = [sharding.RowBlock(Ap.cuda(), Aj.cuda(), w.cuda(), bounds, r) for r in range(world)]
    shards = [sharding.Shard(b, b.Ax) for b in blks]
    assert sum(s.nnz_local_cols + s.nnz_remote_cols for s in shards) == int(Ap[-1])
    # the "exchange": every rank's padded slice, concatenated in rank order = what the all-gather delivers
    slices = [b.pad_local(d[b.lo:b.hi].cuda(), fill=INF) for b in blks]
    v_full = torch.cat(slices)
    got = np.empty(n, np.int32)
    for b, s, v_loc in zip(blks, shards, slices):
        assert s.split_ok and s.serves("INT", "MIN")
        r = torch.full((b.n_local,), -7, dtype=torch.int32, device="cuda")
        m = mask[b.lo:b.hi].cuda()
        s.mxv_local("INT", m, v_loc, r, "PLUS", "MIN", "ALWAYS", INF)
        s.mxv_remote("INT", m, v_full, r, "PLUS", "MIN", "ALWAYS")
        r2 = torch.empty_like(r)
        s.mxv_whole("INT", m, v_full, r2, "PLUS", "MIN", "ALWAYS", INF)
        assert torch.equal(r, r2)
        got[b.lo:b.hi] = r.cpu().numpy()
    np.testing.assert_array_equal(got, want)
    for s in shards:
        s.close()


@pytest.mark.parametrize("case", [("FLOAT", "MULT", "PLUS", "NQZERO", 0.0), ("INT", "LAND", "LOR", "EQZERO", 5), ("UINT", "BAND", "BOR", "ALWAYS", 0),
                                  ("INT", "MULT", "SECOND", "ALWAYS", 0)])
def test_sharded_phases_other_semirings(oracle, case):
    """fp32 plus-times (1e-5), LOR with a non-zero init (rows without an entry keep init itself), BOR; and an op_add the blocks cannot serve
    (SECOND: order-dependent) takes the unsplit block after the exchange."""
    from helpers import NP, random_mask, random_values
    from spla_amd import sharding
    t, mul, add, sel, init = case
    n, Ap, Aj = _graph(12, 8)
    rng = np.random.default_rng(17)
    Ax = random_values(rng, int(Ap[-1]), t)
    v = random_values(rng, n, t, zeros=0.3)
    mask = random_mask(rng, n, t)
    want = oracle.mxv_masked(t, n, n, Ap.numpy(), Aj.numpy(), Ax, mask, v, mul, add, sel, init)
    world = 4
    bounds = sharding.row_splits(Ap, world)
    tx = torch.from_numpy(Ax.view(np.int32) if t == "UINT" else Ax)
    tv = torch.from_numpy(v.view(np.int32) if t == "UINT" else v)
    tm = torch.from_numpy(mask.view(np.int32) if t == "UINT" else mask)
    blks = [sharding.RowBlock(Ap.cuda(), Aj.cuda(), tx.cuda(), bounds, r) for r in range(world)]
    slices = [b.pad_local(tv[b.lo:b.hi].cuda()) for b in blks]
    v_full = torch.cat(slices)
    got = np.empty(n, NP[t])
    for b, v_loc in zip(blks, slices):
        s = sharding.Shard(b, b.Ax)
        r = torch.empty(b.n_local, dtype=tx.dtype, device="cuda")
        m = tm[b.lo:b.hi].cuda()
        if s.serves(t, add):
            s.mxv_local(t, m, v_loc, r, mul, add, sel, init)
            s.mxv_remote(t, m, v_full, r, mul, add, sel)
        else:
            assert add == "SECOND"
            s.mxv_whole(t, m, v_full, r, mul, add, sel, init)
        got[b.lo:b.hi] = r.cpu().numpy().view(NP[t])
        s.close()
    if t == "FLOAT":
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
    else:
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_rccl_communicator_world1_and_dist_product(oracle):
    """splahip_dist_*: RCCL loads, a one-rank communicator initialises, both all-gather forms move the slice, and splahip_dist_mxv_masked (exchange
    stream || local part, then the remote part) returns the oracle's result."""
    from spla_amd import _lib, sharding
    assert _lib.call("splahip_dist_available") == 1
    comm = sharding.DistComm(1, 0)
    x = torch.arange(1000, dtype=torch.int32, device="cuda")
    for direct in (False, True):
        y = torch.zeros_like(x)
        comm.all_gather(x, y, direct=direct)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    n, Ap, Aj = _graph(12, 8)
    g = torch.Generator().manual_seed(5)
    w = torch.randint(1, 50, (int(Ap[-1]),), generator=g, dtype=torch.int32)
    v = torch.randint(0, 100, (n,), generator=g, dtype=torch.int32)
    mask = torch.randint(0, 2, (n,), generator=g, dtype=torch.int32)
    want = oracle.mxv_masked("INT", n, n, Ap.numpy(), Aj.numpy(), w.numpy(), mask.numpy(), v.numpy(), "MULT", "PLUS", "EQZERO", 3)
    blk = sharding.RowBlock(Ap.cuda(), Aj.cuda(), w.cuda(), [0, n], 0)
    s = sharding.Shard(blk, blk.Ax)
    r = torch.empty(n, dtype=torch.int32, device="cuda")
    v_full = torch.empty(n, dtype=torch.int32, device="cuda")
    overlapped = s.product(comm, "INT", mask.cuda(), v.cuda(), v_full, r, "MULT", "PLUS", "EQZERO", 3)
    torch.cuda.synchronize()
    assert overlapped
    np.testing.assert_array_equal(r.cpu().numpy(), want)
    s.close()
    # the unsplit form (what two ranks run by default: the exchange first, then ONE product over the block's own column-ordered blocks), forced here by the switch
    import os
    os.environ["SPLAHIP_DIST_FORM"] = "whole"
    try:
        s2 = sharding.Shard(blk, blk.Ax)
        assert s2.prefers_whole("INT", "PLUS") and s2.serves("INT", "PLUS")
        r.fill_(-5)
        overlapped = s2.product(comm, "INT", mask.cuda(), v.cuda(), v_full, r, "MULT", "PLUS", "EQZERO", 3)
        torch.cuda.synchronize()
        assert not overlapped
        np.testing.assert_array_equal(r.cpu().numpy(), want)
        s2.close()
    finally:
        os.environ.pop("SPLAHIP_DIST_FORM")
    comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 3])
def test_sparse_exchange_c_abi_simulated_ranks(world):
    """splahip_dist_sx_pack / _buffers / _scatter (dist.hip): the sparsity-aware exchange of the input vector, driven rank by rank on one GPU — every
    simulated rank packs its slice, the messages are copied into every rank's receive buffer (what the RCCL all-gather of splahip_dist_exchange_sparse
    does), every rank scatters; each rank's v_full must equal the dense all-gather of the slices, over several rounds (the previous round's positions are
    cleared), with and without values, with a caller-supplied bound, and an overflow must be reported."""
    import ctypes as C
    from spla_amd import _lib, sharding
    n, Ap, Aj = _graph(12)
    bounds = sharding.row_splits(Ap, world)
    blks = [sharding.RowBlock(Ap, Aj, None, bounds, p) for p in range(world)]
    for fill, const in ((0, 1), (7, None)):
        sxs = [sharding.SparseExchangeABI(b, torch.int32, "cuda", fill=fill) for b in blks]
        for rnd in range(5):
            gg = torch.Generator().manual_seed(500 + rnd)
            x = torch.full((n,), fill, dtype=torch.int32)
            k = [n // 40, n // 5, 0, 5, n // 11][rnd]
            pos = torch.randperm(n, generator=gg)[:k]
            x[pos] = 1 if const is not None else torch.randint(8, 99, (k,), generator=gg, dtype=torch.int32)
            xd = x.cuda()
            with_values = const is None
            m = sxs[0].capacity if rnd % 2 == 0 else max(1, min(k, sxs[0].capacity))
            words = 1 + (2 if with_values else 1) * m
            fill_bits = fill
            sends, recvs = [], []
            for p, (b, sx) in enumerate(zip(blks, sxs)):
                xl = xd[b.lo:b.hi].contiguous()
                _lib.call("splahip_dist_sx_pack", sx._h, xl.data_ptr(), int(xl.numel()), fill_bits, m, 1 if with_values else 0, None)
                ps, pr, w = C.c_void_p(), C.c_void_p(), C.c_size_t()
                _lib.call("splahip_dist_sx_buffers", sx._h, m, 1 if with_values else 0, C.byref(ps), C.byref(pr), C.byref(w))
                assert w.value == words
                sends.append(ps.value), recvs.append(pr.value)
            torch.cuda.synchronize()
            for pr in recvs:  # the all-gather: rank q's message lands at word q * words of every rank's receive buffer
                for q, ps in enumerate(sends):
                    _lib.call("splahip_memcpy_d2d", pr + 4 * words * q, ps, 4 * words, None)
            want = torch.full((blks[0].n_cols_padded,), fill, dtype=torch.int32, device="cuda")
            for q, b in enumerate(blks):
                base = q * b.max_rows if world > 1 else 0
                want[base:base + b.n_local] = xd[b.lo:b.hi]
            for sx in sxs:
                _lib.call("splahip_dist_sx_scatter", sx._h, sx.v_full.data_ptr(), fill_bits, m, 1 if with_values else 0, 0 if with_values else const, None)
                torch.cuda.synchronize()
                assert torch.equal(sx.v_full, want), (world, fill, rnd)
                assert not sx.overflowed()
        # a bound smaller than a rank's count: reported
        x = torch.full((n,), fill, dtype=torch.int32, device="cuda")
        x[::3] = 1 if const is not None else 55
        b, sx = blks[0], sxs[0]
        xl = x[b.lo:b.hi].contiguous()
        _lib.call("splahip_dist_sx_pack", sx._h, xl.data_ptr(), int(xl.numel()), fill, 2, 0 if const is not None else 1, None)
        torch.cuda.synchronize()
        assert sx.overflowed()
        for sx in sxs:
            sx.close()
    # the whole exchange through RCCL at world size 1 (one GPU): splahip_dist_exchange_sparse
    if world == 1 and _lib.call("splahip_dist_available"):
        comm = sharding.DistComm(1, 0)
        sx = sharding.SparseExchangeABI(blks[0], torch.float32, "cuda", fill=float("inf"), comm=comm)
        for rnd in range(3):
            x = torch.full((n,), float("inf"), dtype=torch.float32, device="cuda")
            x[torch.randperm(n, generator=torch.Generator().manual_seed(rnd))[: n // (3 + rnd)].cuda()] = float(rnd) + 0.5
            got = sx.exchange(x)
            torch.cuda.synchronize()
            assert torch.equal(got, x)
        sx.close()
        comm.close()


@pytest.mark.gpu
def test_dist_layer_from_cpp_without_python():
    """tests/cpp/test_dist_c_abi.cpp: the decomposition driven through the kernel-level C ABI alone (what a C++ host binds): shards of 1 and
    3 simulated ranks, the dense exchange (RCCL communicator at world size 1 + splahip_dist_mxv_masked), the sparse exchange's pack / scatter kernels, the
    two-part fold — stitched results bit-identical to splahip_mxv_masked on the whole matrix."""
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parents[1] / "oracle" / "_ref" / "test_dist_c_abi"
    if not exe.exists():
        pytest.skip(f"{exe} not built (oracle/Makefile target usertests)")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DIST C-ABI OK" in out.stdout, (out.stdout + out.stderr)[-2000:]


def _lib_trim():
    from spla_amd import _lib
    _lib.call("splahip_mxm_trim")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_spgemm_row_blocks_on_the_hip_kernels(oracle, world):
    """mxm by rows of A, mxmT by rows of the mask (sharding.sharded_mxm / sharded_mxmT): every simulated rank's row block through the HIP
    kernels, concatenated, equals the unsharded HIP call and the oracle — bit-exact for int32; fp32 sums within 1e-5 of the oracle."""
    from spla_amd import graphgen, kernels, sharding
    n, Ap, Aj = graphgen.rmat_csr(11, 8, seed=21, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(22)
    for t, dt, npdt in (("INT", torch.int32, np.int32), ("FLOAT", torch.float32, np.float32)):
        Ax = torch.randint(1, 9, (int(Ap[-1]),), generator=g, device="cuda", dtype=torch.int32).to(dt)
        A = (Ap, Aj, Ax)
        hostA = tuple(x.cpu().numpy() for x in A)
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
        keep = Aj.to(torch.int64) < rows
        Lp = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
        Lp[1:] = torch.cumsum(torch.bincount(rows[keep], minlength=n), 0).to(torch.int32)
        L = (Lp, Aj[keep].contiguous(), Ax[keep].contiguous())
        hostL = tuple(x.cpu().numpy() for x in L)

        def stitch(blocks, bounds):
            base, ps, js, xs = 0, [torch.zeros(1, dtype=torch.int64, device="cuda")], [], []
            for (Rp, Rj, Rx) in blocks:
                ps.append(Rp[1:].to(torch.int64) + base)
                base += int(Rp[-1])
                js.append(Rj)
                xs.append(Rx)
            return torch.cat(ps).cpu().numpy(), torch.cat(js).cpu().numpy().view(np.uint32), torch.cat(xs).cpu().numpy()

        def check(got, want, what):
            assert np.array_equal(got[0], want[0].astype(np.int64)) and np.array_equal(got[1], want[1]), what
            if t == "INT":
                assert np.array_equal(got[2].view(npdt), want[2]), what
            else:
                np.testing.assert_allclose(got[2].view(npdt), want[2], rtol=1e-5, atol=0, err_msg=what)

        blocks = []
        for rank in range(world):
            bounds, local = sharding.sharded_mxm(t, n, n, A, A, "MULT", "PLUS", 0, world, rank)
            blocks.append(local)
        torch.cuda.synchronize()
        check(stitch(blocks, bounds), oracle.mxm(t, n, n, hostA, hostA, "MULT", "PLUS", 0), f"{t} mxm world={world}")
        whole = kernels.mxm(t, n, n, A, A, "MULT", "PLUS", 0)[0]
        if t == "INT":
            got = stitch(blocks, bounds)
            assert np.array_equal(got[1], whole[1].cpu().numpy().view(np.uint32)) and np.array_equal(got[2].view(npdt), whole[2].cpu().numpy())
        blocks, tri = [], 0
        for rank in range(world):
            bounds, local = sharding.sharded_mxmT(t, n, L, L, L, "MULT", "PLUS", "NQZERO", 0, world, rank)
            blocks.append(local)
            tri += float(local[2].to(torch.float64).sum())
        torch.cuda.synchronize()
        want = oracle.mxmT_masked(t, n, hostL, hostL, hostL, "MULT", "PLUS", "NQZERO", 0)
        check(stitch(blocks, bounds), want, f"{t} mxmT world={world}")
        assert abs(tri - float(want[2].astype(np.float64).sum())) <= 1e-6 * max(1.0, tri)
    _lib_trim()
