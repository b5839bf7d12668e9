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
