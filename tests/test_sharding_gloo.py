"""The N > 1 path on CPU: two gloo ranks run the 1-D row-block decomposition (partition, column remap, padded all-gather)
and the row blocks' results (computed by the oracle here — it is the checker) equal the unsharded result."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from spla_amd import graphgen, sharding

    n, Ap, Aj = graphgen.rmat_csr(10, 8, seed=5, device="cpu")
    g = torch.Generator().manual_seed(9)
    Ax = torch.randint(1, 50, (int(Ap[-1]),), generator=g, dtype=torch.int32)
    v = torch.randint(0, 1000, (n,), generator=g, dtype=torch.int32)
    mask = torch.randint(0, 2, (n,), generator=g, dtype=torch.int32)
    bounds = sharding.row_splits(Ap, world)
    blk = sharding.RowBlock(Ap, Aj, Ax, bounds, rank)
    # nnz balance: no block is more than ~2x the mean (scale-free rows make exact balance impossible)
    assert blk.nnz_local <= 2 * int(Ap[-1]) // world + int((Ap[1:] - Ap[:-1]).max())
    # exchange step: all-gather of the padded slices, read through the remapped column ids
    v_pad = blk.all_gather(blk.pad_local(v[blk.lo:blk.hi]))
    assert torch.equal(blk.unpad(v_pad), v)
    assert torch.equal(v_pad[blk.Aj.to(torch.int64)], v[blk.Aj_global.to(torch.int64)])
    # local SpMV over the padded layout == the same rows of the global SpMV
    orc = pyoracle.Oracle()
    r_local = orc.mxv_masked("INT", blk.n_local, blk.n_cols_padded, blk.Ap.numpy(), blk.Aj.numpy(), blk.Ax.numpy(),
                             mask[blk.lo:blk.hi].numpy(), v_pad.numpy(), "PLUS", "MIN", "EQZERO", 1 << 30)
    r_full = orc.mxv_masked("INT", n, n, Ap.numpy(), Aj.numpy(), Ax.numpy(), mask.numpy(), v.numpy(), "PLUS", "MIN", "EQZERO", 1 << 30)
    assert np.array_equal(r_local, r_full[blk.lo:blk.hi])
    # the next iteration's input is the gathered result: results stay sharded, gather again
    r_pad = blk.all_gather(blk.pad_local(torch.from_numpy(r_local)))
    assert np.array_equal(blk.unpad(r_pad).numpy(), r_full)
    # sparsity-aware exchange == dense exchange, over several rounds (the previous round's positions must be cleared), with and
    # without the values travelling, including an all-fill round
    for fill, const in ((0, 1), (7, None)):
        sx = sharding.SparseExchange(blk, torch.int32, "cpu", fill=fill)
        for rnd in range(4):
            gg = torch.Generator().manual_seed(100 + rnd)
            x = torch.full((n,), fill, dtype=torch.int32)
            k = [n // 50, n // 7, 0, 3][rnd]
            pos = torch.randperm(n, generator=gg)[:k]
            x[pos] = 1 if const is not None else torch.randint(8, 99, (k,), generator=gg, dtype=torch.int32)
            # odd rounds: the caller knows a global bound on the count (no count exchange, no host sync); even rounds: it does not
            got = sx.exchange(x[blk.lo:blk.hi], value=const, upper_bound=(k if rnd % 2 else None))
            want = blk.all_gather(blk.pad_local(x[blk.lo:blk.hi], fill=fill))
            assert got.numel() == blk.n_cols_padded + 1
            live = torch.zeros(blk.n_cols_padded, dtype=torch.bool)  # padding slots of the dense layout hold `fill` in both forms
            assert torch.equal(got[:blk.n_cols_padded], want), (fill, rnd)
            assert torch.equal(got[:blk.n_cols_padded][blk.Aj.to(torch.int64)], x[blk.Aj_global.to(torch.int64)])
            assert sx.last_count == (None if rnd % 2 else k)
    # the same exchange in the C ABI's protocol (splahip_dist_exchange_sparse: a message = [count | m positions | m values], m = a bound every rank
    # knows; two receive buffers alternate, the older one lists what to clear).  On CPU tensors SparseExchangeABI runs the word-for-word host mirror
    # of the pack / scatter kernels (this container has no GPU; tests/test_shard_gpu.py drives the kernels themselves), over the gloo all-gather
    for fill, const in ((0, 1), (7, None)):
        sa = sharding.SparseExchangeABI(blk, torch.int32, "cpu", fill=fill, capacity=max(1, blk.max_rows if world > 1 else n))
        for rnd in range(5):
            gg = torch.Generator().manual_seed(300 + rnd)
            x = torch.full((n,), fill, dtype=torch.int32)
            k = [n // 50, n // 7, 0, 3, n // 9][rnd]
            pos = torch.randperm(n, generator=gg)[:k]
            x[pos] = 1 if const is not None else torch.randint(8, 99, (k,), generator=gg, dtype=torch.int32)
            got = sa.exchange(x[blk.lo:blk.hi], value=const, upper_bound=(max(k, 1) if rnd % 2 else None))
            want = blk.all_gather(blk.pad_local(x[blk.lo:blk.hi], fill=fill))
            assert torch.equal(got, want), ("abi protocol", fill, rnd)
            assert not sa.overflowed()
            assert sa.last_words == 1 + (1 if const is not None else 2) * (min(max(k, 1), sa.capacity) if rnd % 2 else sa.capacity)
        # a bound that is too small is reported, not silently wrong
        x = torch.full((n,), fill, dtype=torch.int32)
        x[:: 3] = 1 if const is not None else 55
        sa.exchange(x[blk.lo:blk.hi], value=const, upper_bound=2)
        assert sa.overflowed()
    # the two-part (overlappable) form of the product: the block cut by column owner; the local part needs only this rank's slice of v,
    # the remote part folds onto its result once the exchange has landed (splahip_shard_mxv_local / _remote, dist.hip).  BASELINE configs[3]:
    # min-plus int32, init = INF = 2^30, weights 1..255; and plus-times / or-and for the other monoid families
    (LAp, LAj, LAx), (RAp, RAj, RAx) = sharding.split_columns(blk)
    assert int(LAp[-1]) + int(RAp[-1]) == blk.nnz_local and (world == 1 or int(RAp[-1]) > 0)
    w = torch.randint(1, 256, (int(Ap[-1]),), generator=g, dtype=torch.int32)
    wblk = sharding.RowBlock(Ap, Aj, w, bounds, rank)
    (_, _, LAw), (_, _, RAw) = sharding.split_columns(wblk)
    INF = 1 << 30
    d = torch.full((n,), INF, dtype=torch.int32)
    d[torch.randperm(n, generator=g)[: n // 8]] = torch.randint(0, 5000, (n // 8,), generator=g, dtype=torch.int32)
    d_pad = blk.all_gather(blk.pad_local(d[blk.lo:blk.hi], fill=INF))
    v_loc = blk.pad_local(d[blk.lo:blk.hi], fill=INF)
    for (mul, add, init, LX, RX, vec_l, vec_f) in (("PLUS", "MIN", INF, LAw, RAw, v_loc, d_pad),
                                                    ("MULT", "PLUS", 0, LAx, RAx, blk.pad_local(v[blk.lo:blk.hi]), v_pad),
                                                    ("LAND", "LOR", 0, LAx, RAx, blk.pad_local(v[blk.lo:blk.hi]), v_pad)):
        whole_x = wblk.Ax if mul == "PLUS" else blk.Ax
        whole_v = d if mul == "PLUS" else v
        want = orc.mxv_masked("INT", n, n, Ap.numpy(), Aj.numpy(), (w if mul == "PLUS" else Ax).numpy(), mask.numpy(), whole_v.numpy(), mul, add, "EQZERO", init)[blk.lo:blk.hi]
        m_loc = mask[blk.lo:blk.hi].numpy()
        r1 = orc.mxv_masked("INT", blk.n_local, blk.max_rows, LAp.numpy(), LAj.numpy(), LX.numpy(), m_loc, vec_l.numpy(), mul, add, "EQZERO", init)
        # remote part: fold from the monoid's identity, then fold onto r1 where the row is selected and holds a remote entry
        ident = {"MIN": 2 ** 31 - 1, "PLUS": 0, "LOR": 0}[add]
        r2 = orc.mxv_masked("INT", blk.n_local, blk.n_cols_padded, RAp.numpy(), RAj.numpy(), RX.numpy(), m_loc, vec_f.numpy(), mul, add, "EQZERO", ident)
        has = (np.diff(RAp.numpy().astype(np.int64)) > 0) & (m_loc == 0)
        fold = {"MIN": np.minimum, "PLUS": np.add, "LOR": lambda a, b: ((a != 0) | (b != 0)).astype(np.int32)}[add]  # reference src/op.cpp:194-234
        got = np.where(has, fold(r1, r2), r1)
        assert np.array_equal(got, want), (mul, add)
        whole = orc.mxv_masked("INT", blk.n_local, blk.n_cols_padded, blk.Ap.numpy(), blk.Aj.numpy(), whole_x.numpy(), m_loc, vec_f.numpy(), mul, add, "EQZERO", init)
        assert np.array_equal(whole, want)
    Path(out_dir, f"ok{rank}").write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_block_decomposition_gloo(tmp_path, world):
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_splits_balance_nnz():
    sys.path.insert(0, str(ROOT))
    from spla_amd import graphgen
    n, Ap, Aj = graphgen.rmat_csr(12, 16, seed=2, device="cpu")
    for parts in (2, 4, 8):
        b = graphgen.nnz_balanced_row_splits(Ap, parts)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(parts))
        nnz = [int(Ap[b[i + 1]] - Ap[b[i]]) for i in range(parts)]
        assert max(nnz) <= int(Ap[-1]) // parts + int((Ap[1:] - Ap[:-1]).max())


def test_owner_sorted_permutation():
    """bench.py's relabelling for N > 1: a bijection whose equal blocks each hold a pseudo-random share of the vertices in original order"""
    sys.path.insert(0, str(ROOT))
    from spla_amd import graphgen
    n, owners = 1 << 12, 8
    perm = graphgen.owner_sorted_permutation(n, owners, seed=3)
    assert sorted(perm.tolist()) == list(range(n))
    inv = torch.empty(n, dtype=torch.int64)
    inv[perm] = torch.arange(n)
    for b in range(owners):
        block = inv[b * n // owners:(b + 1) * n // owners]
        assert torch.all(block[1:] > block[:-1])  # original order inside a block
        assert abs(float(block.double().mean()) - n / 2) < n / 8  # a spread sample of the id space, not a contiguous range
    # the relabelled graph is the same graph
    _, Ap0, Aj0 = graphgen.rmat_csr(10, 8, seed=4, device="cpu")
    _, Ap1, Aj1 = graphgen.rmat_csr(10, 8, seed=4, device="cpu", permute="owner-sorted", owners=4)
    assert int(Ap0[-1]) == int(Ap1[-1])
    assert sorted((Ap0[1:] - Ap0[:-1]).tolist()) == sorted((Ap1[1:] - Ap1[:-1]).tolist())


def _spgemm_worker(rank, world, port, out_dir):
    """mxm by rows of A and mxmT by rows of the mask, B whole on every rank: the gathered row blocks are the unsharded result word for word.
    The local products are the oracle's here (the checker; this container has no GPU) — tests/test_shard_gpu.py runs the same driver on the HIP kernels."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from spla_amd import graphgen, sharding
    orc = pyoracle.Oracle()

    def as_t(triple):
        return tuple(torch.from_numpy(np.ascontiguousarray(x).view(np.int32) if x.dtype == np.uint32 else np.ascontiguousarray(x)) for x in triple)

    def o_mxm(t, a_rows, b_cols, A, B, op_mult, op_add, init):
        return as_t(orc.mxm(t, a_rows, b_cols, tuple(x.numpy() for x in A), tuple(x.numpy() for x in B), op_mult, op_add, init))

    def o_mxmT(t, n_rows, M, A, B, op_mult, op_add, op_select, init):
        return as_t(orc.mxmT_masked(t, n_rows, tuple(x.numpy() for x in M), tuple(x.numpy() for x in A), tuple(x.numpy() for x in B), op_mult, op_add, op_select, init))

    n, Ap, Aj = graphgen.rmat_csr(9, 8, seed=11, device="cpu")
    g = torch.Generator().manual_seed(12)
    for t, dt in (("INT", torch.int32), ("FLOAT", torch.float32)):
        Ax = torch.randint(1, 9, (int(Ap[-1]),), generator=g, dtype=torch.int32).to(dt)
        A = (Ap, Aj, Ax)
        # mxm: the blocks balance the products; every rank's rows equal the same rows of the whole product
        want = o_mxm(t, n, n, A, A, "MULT", "PLUS", 0)
        bounds, local = sharding.sharded_mxm(t, n, n, A, A, "MULT", "PLUS", 0, world, rank, product=o_mxm)
        cost = sharding.mxm_row_cost(Ap, Aj, Ap)
        share = [int(cost[bounds[p]:bounds[p + 1]].sum()) for p in range(world)]
        assert bounds[0] == 0 and bounds[-1] == n and max(share) <= int(cost.sum()) // world + int(cost.max())
        got = sharding.gather_csr_rows(local, bounds)
        for a, b in zip(got, want):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (t, "mxm", world)
        lo, hi = bounds[rank], bounds[rank + 1]
        assert torch.equal(local[0].to(torch.int64), want[0][lo:hi + 1].to(torch.int64) - int(want[0][lo]))
        # mxmT: the strictly lower triangle L as mask, A and B (the triangle-count call, reference src/algorithm.cpp:380-414), and a general mask
        rows = torch.repeat_interleave(torch.arange(n), (Ap[1:] - Ap[:-1]).to(torch.int64))
        keep = Aj.to(torch.int64) < rows
        Lp = torch.zeros(n + 1, dtype=torch.int32)
        Lp[1:] = torch.cumsum(torch.bincount(rows[keep], minlength=n), 0).to(torch.int32)
        L = (Lp, Aj[keep].contiguous(), Ax[keep].contiguous())
        for (M, X, Y, sel) in ((L, L, L, "NQZERO"), (A, L, A, "GTZERO")):
            want = o_mxmT(t, n, M, X, Y, "MULT", "PLUS", sel, 0)
            bounds, local = sharding.sharded_mxmT(t, n, M, X, Y, "MULT", "PLUS", sel, 0, world, rank, product=o_mxmT)
            got = sharding.gather_csr_rows(local, bounds)
            for a, b in zip(got, want):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (t, "mxmT", sel, world)
            # the triangle count is one all-reduced scalar of the local sums (no gather of R in the data path)
            if M is L and t == "INT":
                part = torch.tensor([int(local[2].to(torch.int64).sum())])
                dist.all_reduce(part)
                assert int(part[0]) == int(want[2].to(torch.int64).sum())
    # empty blocks (more ranks than rows with work): a 3-row matrix whose only entries sit in row 1
    Ep = torch.tensor([0, 0, 2, 2], dtype=torch.int32)
    Ej = torch.tensor([0, 2], dtype=torch.int32)
    Ex = torch.tensor([3, 4], dtype=torch.int32)
    E = (Ep, Ej, Ex)
    want = o_mxm("INT", 3, 3, E, E, "MULT", "PLUS", 0)
    bounds, local = sharding.sharded_mxm("INT", 3, 3, E, E, "MULT", "PLUS", 0, world, rank, product=o_mxm)
    got = sharding.gather_csr_rows(local, bounds)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    Path(out_dir, f"ok{rank}").write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_spgemm_gloo(tmp_path, world):
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_spgemm_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
