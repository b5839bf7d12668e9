"""bench.py's N > 1 products through the kernel-level C ABI (spla_amd/csrc/kernels/dist.hip): one process per GPU, 1-D nnz-balanced row blocks,
RCCL all-gather of the input vector on the communicator's stream overlapped with the local-column part of the SpMV, the remote-column part
folding onto it (splahip_dist_mxv_masked).

    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --workload sssp-step
        BASELINE configs[3]: exec_mxv min-plus int32 on R-MAT scale 24, STRONG split (the global matrix is fixed, every rank owns 1/N of its nnz).

SPLA_DIST_BACKEND=gloo (debug): ranks may share one GPU; RCCL cannot host two ranks on one device, so the exchange is staged through the host
with torch.distributed while the phases of the product (local part / remote part) still run through the C ABI — the overlap of the exchange with
the local part is then the overlap of a CPU-side collective with a GPU kernel.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0


def _log(*a):
    import sys
    print(*a, file=sys.stderr, flush=True)


class ShardedProduct:
    """this rank's shard + the exchange; step() = one sharded masked SpMV"""

    def __init__(self, blk, Ax, t, mul, add, sel, init, fill, world, rank, direct=False):
        from spla_amd import sharding
        self.blk, self.t, self.ops, self.init, self.world, self.rank, self.direct = blk, t, (mul, add, sel), init, world, rank, direct
        self.shard = sharding.Shard(blk, Ax)
        self.two_part = self.shard.serves(t, add) and not self.shard.prefers_whole(t, add)  # (what splahip_dist_mxv_masked itself decides)
        self.rccl = world > 1 and dist.get_backend() == "nccl"
        self.comm = None
        self.fallback = None
        if self.rccl:
            try:
                self.comm = sharding.DistComm(world, rank)
            except Exception as exc:  # the torch.distributed exchange stays available
                self.fallback = f"splahip_dist_comm_create failed ({exc}); exchange through torch.distributed"
                self.rccl = False
        dt = torch.int32 if t != "FLOAT" else torch.float32
        self.v_full = torch.full((blk.n_cols_padded,), fill, dtype=dt, device="cuda")
        self.ev = (torch.cuda.Event(), torch.cuda.Event())

    def exchange(self, v_local_padded):
        if self.world == 1:
            self.v_full.copy_(v_local_padded)
        elif self.rccl:
            self.comm.all_gather(v_local_padded, self.v_full, direct=self.direct)
        else:
            self.blk.all_gather(v_local_padded, self.v_full)

    def step(self, mask, v_local_padded, r):
        mul, add, sel = self.ops
        if self.rccl:
            self.shard.product(self.comm, self.t, mask, v_local_padded, self.v_full, r, mul, add, sel, self.init, direct=self.direct)
        elif self.two_part:
            self.shard.mxv_local(self.t, mask, v_local_padded, r, mul, add, sel, self.init)  # queued on the GPU ...
            self.exchange(v_local_padded)                                                      # ... while the host-staged exchange runs
            self.shard.mxv_remote(self.t, mask, self.v_full, r, mul, add, sel)
        else:
            self.exchange(v_local_padded)
            self.shard.mxv_whole(self.t, mask, self.v_full, r, mul, add, sel, self.init)

    def close(self):
        self.shard.close()
        if self.comm:
            self.comm.close()


def timed(fn, steps, warmup, world):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        w = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w[0])
    return wall, [a.elapsed_time(b) for a, b in evs]


def phase_report(sp: ShardedProduct, mask, v_loc, r, steps, world):
    """exchange / local part / remote part timed on their own, and the overlap the whole step achieves:
    overlap_fraction = (t_exchange + t_local + t_remote - t_step) / min(t_exchange, t_local), clipped to [0, 1]"""
    mul, add, sel = sp.ops
    _, px = timed(lambda: sp.exchange(v_loc), steps, 2, world)
    out = {"exchange_avg_us": round(float(np.mean(px)) * 1e3, 2)}
    if sp.two_part:
        _, pl = timed(lambda: sp.shard.mxv_local(sp.t, mask, v_loc, r, mul, add, sel, sp.init), steps, 2, world)
        _, pr = timed(lambda: sp.shard.mxv_remote(sp.t, mask, sp.v_full, r, mul, add, sel), steps, 2, world)
        out["local_part_avg_us"] = round(float(np.mean(pl)) * 1e3, 2)
        out["remote_part_avg_us"] = round(float(np.mean(pr)) * 1e3, 2)
    else:  # the unsplit block over the gathered vector (what two ranks run): the product on its own
        _, pw = timed(lambda: sp.shard.mxv_whole(sp.t, mask, sp.v_full, r, mul, add, sel, sp.init), steps, 2, world)
        out["whole_part_avg_us"] = round(float(np.mean(pw)) * 1e3, 2)
    _, ps = timed(lambda: sp.step(mask, v_loc, r), steps, 2, world)
    out["step_avg_us"] = round(float(np.mean(ps)) * 1e3, 2)
    if sp.two_part and world > 1:
        serial = out["exchange_avg_us"] + out["local_part_avg_us"] + out["remote_part_avg_us"]
        hide = min(out["exchange_avg_us"], out["local_part_avg_us"])
        out["overlap_fraction"] = round(max(0.0, min(1.0, (serial - out["step_avg_us"]) / hide)), 3) if hide > 0 else 0.0
    else:
        out["overlap_fraction"] = 0.0
    return out


def sssp_step_sharded(args, rank, world):
    """BASELINE configs[3]: exec_mxv min-plus int32 on R-MAT scale 24, 1-D row split across N GPUs with the all-gather of the distance vector."""
    from oracle import pyoracle
    from spla_amd import graphgen, sharding
    scale = 24 if args.scale == 22 else args.scale
    pm = args.permute if args.permute >= 0 else 2
    permute = {0: False, 1: True, 2: "owner-sorted"}[pm]
    t0 = time.time()
    n, Ap, Aj = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda", permute=permute, owners=args.owners or world)
    nnz = int(Ap[-1])
    g = torch.Generator(device="cuda")
    g.manual_seed(args.seed + 5)  # every rank draws the same global weights / distances
    Ax = torch.randint(1, 256, (nnz,), generator=g, device="cuda", dtype=torch.int32)
    INF = 1 << 30
    v = torch.full((n,), INF, dtype=torch.int32, device="cuda")
    idx = torch.randint(0, n, (n // 8,), generator=g, device="cuda")
    v[idx] = torch.randint(0, 5000, (n // 8,), generator=g, device="cuda", dtype=torch.int32)
    bounds = sharding.row_splits(Ap, world)
    blk = sharding.RowBlock(Ap, Aj, Ax, bounds, rank)
    del Ap, Aj, Ax
    torch.cuda.empty_cache()
    if rank == 0:
        _log(f"[bench] sssp-step x{world}: R-MAT scale {scale} ({permute} labels) n={n} nnz={nnz} gen {time.time() - t0:.1f}s; row splits {bounds}")
    sp = ShardedProduct(blk, blk.Ax, "INT", "PLUS", "MIN", "ALWAYS", INF, INF, world, rank, direct=os.environ.get("SPLA_ALLGATHER", "nccl") == "direct")
    v_loc = blk.pad_local(v[blk.lo:blk.hi], fill=INF)
    mask = torch.zeros(blk.n_local, dtype=torch.int32, device="cuda")
    r = torch.empty(blk.n_local, dtype=torch.int32, device="cuda")
    wall, per = timed(lambda: sp.step(mask, v_loc, r), args.steps, args.warmup, world)
    ms = wall / args.steps * 1e3
    phases = phase_report(sp, mask, v_loc, r, max(5, args.steps // 2), world)
    sp.step(mask, v_loc, r)
    torch.cuda.synchronize()
    # verification: every 64th local row against the oracle on the gathered vector
    ok = True
    if not args.no_verify:
        orc = pyoracle.Oracle()
        rows = torch.arange(0, blk.n_local, 64, device="cuda")
        Ap64 = blk.Ap.to(torch.int64)
        deg = (Ap64[1:] - Ap64[:-1])[rows]
        Aps = torch.zeros(rows.numel() + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(deg, 0, out=Aps[1:])
        rep = torch.repeat_interleave(torch.arange(rows.numel(), device="cuda"), deg)
        src = Ap64[rows][rep] + (torch.arange(int(Aps[-1]), device="cuda") - Aps[:-1][rep])
        want = orc.mxv_masked("INT", int(rows.numel()), blk.n_cols_padded, Aps.to(torch.int32).cpu().numpy().view(np.uint32), blk.Aj[src].cpu().numpy().view(np.uint32),
                              blk.Ax[src].cpu().numpy(), np.zeros(int(rows.numel()), np.int32), sp.v_full.cpu().numpy(), "PLUS", "MIN", "ALWAYS", INF)
        ok = bool(np.array_equal(want, r[rows].cpu().numpy()))
        flag = torch.tensor([1 if ok else 0], device="cpu" if world > 1 and dist.get_backend() == "gloo" else "cuda")
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(int(flag[0]))
        if not ok:
            raise SystemExit("bench: sharded HIP result differs from the oracle")
    # roofline of this rank's launch: algorithmic bytes of its row block / the step's kernel time (local + remote parts)
    k_us = phases.get("local_part_avg_us", 0.0) + phases.get("remote_part_avg_us", 0.0) if sp.two_part else phases.get("whole_part_avg_us", phases["step_avg_us"] - phases["exchange_avg_us"])
    alg = 8 * blk.nnz_local + 4 * (blk.n_local + 1) + 8 * blk.n_local + 4 * blk.n_cols_padded
    gbs = alg / (k_us * 1e-6) / 1e9 if k_us > 0 else 0.0
    slice_bytes = 4 * blk.max_rows
    line = {"metric": "masked SpMV GTEPS (SSSP step)", "value": round(nnz / (ms * 1e-3) / 1e9, 3), "unit": "GTEPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"exec_mxv_masked PLUS/MIN int32 (min-plus), ALWAYS, init 2^30, weights 1..255, 1/8 of v finite; R-MAT scale {scale} ef {args.edge_factor} "
                                   f"({permute} labels), 1-D nnz-balanced row blocks x{world}, all-gather of the distance vector per step", "n": n, "nnz": nnz,
                       "nnz_rank0": blk.nnz_local, "rows_rank0": blk.n_local, "parallelism": f"1-D row blocks x{world} + RCCL all-gather(v) overlapped with the local-column part"},
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "mxv_cob (local-column part) + mxv_cob accumulate (remote-column part)" if sp.two_part else "unsplit block", "kernel_avg_us": round(k_us, 2),
                         "algorithmic_bytes": alg, "note": "rank 0's row block"},
            "exchange": {"mode": "splahip_dist_mxv_masked: RCCL " + ("grouped send/recv per peer (direct)" if sp.direct else "ncclAllGather") + " on the communicator's stream"
                                 if sp.rccl else "torch.distributed (" + (dist.get_backend() if world > 1 else "none") + ") between the C-ABI phases",
                         "bytes_per_peer": slice_bytes, "bytes_received_per_rank": slice_bytes * (world - 1), "two_part_form": sp.two_part,
                         "gbs_per_link_if_direct": round(slice_bytes / (phases["exchange_avg_us"] * 1e-6) / 1e9, 2) if phases["exchange_avg_us"] else None, **phases},
            "shard": {"nnz_local_columns": sp.shard.nnz_local_cols, "nnz_remote_columns": sp.shard.nnz_remote_cols}, "verified": ok}
    if sp.fallback:
        line["exchange"]["fallback"] = sp.fallback
    sp.close()
    return line


def _lib_trim():
    """give the SpGEMM's cached scratch back (splahip_mxm_trim)"""
    from spla_amd import _lib
    _lib.call("splahip_mxm_trim")


def spgemm_sharded(args, rank, world):
    """BASELINE configs[4] across N GPUs: the triangle-count call mxmT(L, L, L) by rows of the mask, or A x A by rows of A; B whole on every
    rank, no exchange in the data path (sharding.sharded_mxmT / sharded_mxm).  STRONG scaling: the matrix is fixed, the row blocks balance the merge steps /
    products.  The only collective is the reduction of one scalar per rank (the triangle count / the result's entry count), outside the timed region."""
    from oracle import pyoracle
    from spla_amd import graphgen, kernels, sharding
    import bench_workloads as bw
    tc = args.workload == "tc-mxmT"
    scale = (20 if tc else 18) if args.scale == 22 else args.scale
    dev = "cpu" if world > 1 and dist.get_backend() == "gloo" else "cuda"
    if tc:
        n, Ap, Aj = bw._sym_rmat(scale, args.edge_factor, args.seed)
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
        low = Aj.to(torch.int64) < rows
        Lp, Lj = graphgen.keys_to_csr(rows[low] * n + Aj[low].to(torch.int64), n, n)
        del rows, low, Ap, Aj
        A = (Lp, Lj, torch.ones(int(Lp[-1]), dtype=torch.int32, device="cuda"))
        cost = sharding.mxmT_row_cost(A[0], A[1], A[0], A[0])
        t = "INT"
    else:
        n, Ap, Aj = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda")
        g = torch.Generator(device="cuda")
        g.manual_seed(args.seed + 9)
        A = (Ap, Aj, (torch.rand(int(Ap[-1]), generator=g, device="cuda") * 0.5 + 0.25).contiguous())
        cost = sharding.mxm_row_cost(A[0], A[1], A[0])
        t = "FLOAT"
    nnz, work = int(A[0][-1]), int(cost.sum())
    bounds = sharding.work_splits(cost, world)
    my_work = int(cost[bounds[rank]:bounds[rank + 1]].sum())
    if rank == 0:
        _log(f"[bench] {args.workload} x{world}: R-MAT scale {scale}: n={n} nnz={nnz} work={work}; row blocks {bounds}")
    out = {}

    def step():
        if tc:
            out["R"] = sharding.sharded_mxmT(t, n, A, A, A, "MULT", "PLUS", "GTZERO", 0, world, rank, bounds=bounds)[1]
        else:
            out["R"] = sharding.sharded_mxm(t, n, n, A, A, "MULT", "PLUS", 0.0, world, rank, bounds=bounds)[1]

    steps = args.steps if tc else max(1, min(args.steps, 5))
    wall, per = timed(step, steps, 1 if not tc else args.warmup, world)
    ms = wall / steps * 1e3
    Rp, Rj, Rx = out["R"]
    tally = torch.tensor([int(Rx.to(torch.int64).sum()) if tc else 0, int(Rj.numel()), my_work], dtype=torch.int64, device=dev)
    shares = [torch.zeros_like(tally) for _ in range(world)]
    if world > 1:
        dist.all_gather(shares, tally)
    else:
        shares = [tally]
    total = torch.stack(shares).sum(0)
    ok = None
    if not args.no_verify:  # every 64th row of this rank's block against the oracle (mask / A restricted to the sample; B whole)
        lo, hi = bounds[rank], bounds[rank + 1]
        srows = torch.arange(lo, hi, 64, device="cuda")
        Ms = bw._row_sample_csr(A[0], A[1], A[2], srows) if tc else None
        Ah = tuple(x.cpu().numpy() for x in A)
        if tc:
            got = kernels.mxmT_masked(t, n, Ms, A, A, "MULT", "PLUS", "GTZERO", 0)
            want = pyoracle.Oracle().mxmT_masked(t, n, tuple(x.cpu().numpy() for x in Ms), Ah, Ah, "MULT", "PLUS", "GTZERO", 0)
            # (the sampled mask keeps the rows' global positions: rows outside the sample are empty) and the block's own rows agree with it
            ok = all(np.array_equal(np.asarray(a).view(np.uint32), b.cpu().numpy().view(np.uint32)) for a, b in zip(want, got))
            sel = (srows - lo).to(torch.int64)
            rp = Rp.to(torch.int64)
            ok = ok and bool(torch.equal((rp[1:] - rp[:-1])[sel], (got[0].to(torch.int64)[1:] - got[0].to(torch.int64)[:-1])[srows.to(torch.int64)]))
        else:
            sub = sharding.CsrRows(A, lo, min(hi, lo + 256))  # the block's first 256 rows
            want = pyoracle.Oracle().mxm(t, sub.n_rows, n, tuple(x.cpu().numpy() for x in sub.triple), Ah, "MULT", "PLUS", 0.0)
            e = int(Rp[sub.n_rows])
            ok = bool(np.array_equal(np.asarray(want[0]).astype(np.int64), Rp[:sub.n_rows + 1].cpu().numpy().astype(np.int64))) and \
                bool(np.array_equal(np.asarray(want[1]).view(np.uint32), Rj[:e].cpu().numpy().view(np.uint32))) and \
                bool(np.allclose(np.asarray(want[2]), Rx[:e].cpu().numpy(), rtol=1e-5, atol=0))
        flag = torch.tensor([1 if ok else 0], device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(int(flag[0]))
        if not ok:
            raise SystemExit(f"bench: sharded {args.workload} differs from the oracle")
    nnz_r = int(total[1])
    alg = 8 * (3 * nnz + nnz_r) + 16 * (n + 1) if tc else 8 * (2 * nnz + nnz_r) + 12 * (n + 1)
    gbs = alg / (ms * 1e-3) / 1e9
    unit = "Gsteps/s" if tc else "Gproducts/s"
    line = {"metric": "masked SpGEMM (mxmT) G merge-steps/s" if tc else "SpGEMM (mxm) G products/s", "value": round(work / (ms * 1e-3) / 1e9, 3), "unit": unit, "n_gpus": world,
            "steps": steps, "warmup": args.warmup if tc else 1, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32" if tc else "f32", "data": "synthetic",
            "config": {"workload": (f"exec_mxmT_masked(L, L, L) MULT/PLUS/GTZERO int32 (triangle-count form), L = strict lower triangle of symmetrised R-MAT scale {scale}" if tc else
                                    f"exec_mxm A x A MULT/PLUS fp32, R-MAT scale {scale}") + f" ef {args.edge_factor}; rows of the mask / of A in {world} work-balanced blocks, B whole on every rank",
                       "n": n, "nnz": nnz, "work": work, "nnz_R": nnz_r, "row_blocks": bounds, "work_share_per_rank": [int(s[2]) for s in shares],
                       **({"triangles": int(total[0])} if tc else {})},
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0 * world, "unit": "GB/s", "frac": round(gbs / (8000.0 * world), 4), "traffic": None,
                         "kernel": "mxmT_entry + mxmT_heavy + compaction" if tc else "mxm_spa row classes", "algorithmic_bytes": int(alg),
                         "note": "whole-job algorithmic bytes (operands counted once) / the slowest rank's call time, against N x 8 TB/s"},
            "rank0_call_ms": [round(x, 3) for x in per[:5]]}
    if ok is not None:
        line["verified"] = ok
    _lib_trim()
    return line
