#!/usr/bin/env python3
"""bench.py — masked SpMV (exec_mxv_masked hot path) on MI355X: GTEPS + achieved HBM GB/s vs roofline.

Default workload (`--workload spmv-fp32`, the contract line since round 3) = the configuration north_star's >= 60 % target is quoted on:
fp32 MULT / PLUS over all rows of a ~100 M-nnz scale-free CSR (R-MAT scale 23, 131 M nnz) — bench_spmv.py.  A "step" is ONE masked SpMV with
the matrix (and its column-ordered-blocks copy) and the vectors resident in HBM.  With N > 1 ranks (one process per GPU, launched by
torch.distributed.run) the matrix is the R-MAT of scale 23 + log2(N) split into N nnz-balanced 1-D row blocks (weak scaling: ~131 M nnz per
GPU) and every step runs through the C-ABI dist layer (splahip_dist_mxv_masked: RCCL all-gather of the dense vector overlapped with the
local-column part).

`--workload bfs-step` = BASELINE.json configs[1], the contract line of rounds 1-2 (kept as variants[0] of the default line): one masked SpMV
of the BFS step on R-MAT scale 22, int32, op_mult = BAND, op_add = BOR, mask = BFS depth vector at the peak-frontier level with EQZERO
(selects unvisited rows), v = that level's frontier as a dense 0/1 vector, init 0.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`, and a `variants` list with the other measured
forms of the same kernel family.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def coll_device():
    """device for the small control-plane collectives (cpu under the gloo debug backend)"""
    return "cpu" if dist.is_initialized() and dist.get_backend() == "gloo" else "cuda"


def kernel_name(plan, t, mul, add, early):
    from spla_amd import kernels
    return kernels.mxv_variant(t, mul, add, early, plan)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--scale", type=int, default=0, help="R-MAT scale per GPU (global scale = scale + log2(gpus)); 0 = the workload's own (23 for spmv-fp32, 22 for bfs-step, ...)")
    p.add_argument("--edge-factor", type=int, default=16)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--permute", type=int, default=-1, help="vertex relabelling: 0 raw, 1 random, 2 owner-sorted (one pseudo-random share of the "
                   "vertices per row block, original order inside a block); default: raw at 1 GPU, owner-sorted when gpus > 1")
    p.add_argument("--owners", type=int, default=0, help="blocks of the owner-sorted relabelling (default: the number of GPUs)")
    p.add_argument("--cpu-sample-nnz", type=int, default=8_000_000, help="nnz of the leading row block timed on the CPU")
    p.add_argument("--plan", default="sell", choices=["sell", "csr"],
                   help="matrix layout of the timed step: sell = CSR + the sliced-ELL copy (splahip_csr_plan_attach_sell, built once per matrix, "
                        "+12 B per entry); csr = the CSR wave-tile kernels only")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-variants", action="store_true")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--no-scale23", action="store_true", help="skip the R-MAT scale-23 fp32 plus-times variant (the ~100 M-nnz north-star configuration)")
    p.add_argument("--workload", default="spmv-fp32", choices=["spmv-fp32", "bfs-step", "pagerank-vxm", "vxm-push", "sssp-step", "tc-mxmT", "mxm", "mxm-19", "mxm-dense", "algorithms", "transpose"],
                   help="spmv-fp32 (default, the contract line: the north-star configuration), bfs-step (BASELINE configs[1], the contract line of rounds 1-2) "
                        "or one of the other hot-path operators (bench_workloads.py; N = 1 only)")
    a = p.parse_args()
    a.scale_given = a.scale != 0
    if a.scale == 0 and a.workload != "spmv-fp32":
        a.scale = 22  # (bench_workloads.py maps 22 to each workload's own scale)
    return a


# ---------------------------------------------------------------------------------------------------------------
# workload construction (all on the GPU, seeded)
# ---------------------------------------------------------------------------------------------------------------
def build_graph(args, rank, world):
    """Seeded R-MAT on the GPU; every rank generates the same global graph and keeps its nnz-balanced row block."""
    from spla_amd import graphgen, sharding
    gscale = args.scale + int(round(math.log2(world)))
    pm = (2 if world > 1 else 0) if args.permute < 0 else args.permute
    permute = {0: False, 1: True, 2: "owner-sorted"}[pm]
    t0 = time.time()
    n, Ap, Aj = graphgen.rmat_csr(gscale, args.edge_factor, seed=args.seed, device="cuda", permute=permute, owners=args.owners or world)
    torch.cuda.synchronize()
    stats = graphgen.degree_stats(Ap)
    bounds = sharding.row_splits(Ap, world)
    blk = sharding.RowBlock(Ap, Aj, None, bounds, rank)
    nnz_global = int(Ap[-1])
    del Ap, Aj
    torch.cuda.empty_cache()
    if rank == 0:
        log(f"[bench] R-MAT scale {gscale} permute={permute}: n={n} nnz={nnz_global} deg min/avg/sd/max="
            f"{stats['min']:.0f}/{stats['avg']:.2f}/{stats['sd']:.1f}/{stats['max']:.0f} empty_rows={stats['empty_rows']} "
            f"gen {time.time() - t0:.1f}s; row splits {bounds}")
    return dict(n=n, gscale=gscale, permute=permute, bounds=bounds, lo=blk.lo, hi=blk.hi, Ap=blk.Ap, Aj=blk.Aj, blk=blk,
                nnz_local=blk.nnz_local, nnz_global=nnz_global, stats=stats)


class Sharded:
    """This rank's row block (spla_amd.sharding.RowBlock: column ids already remapped to the padded all-gather layout),
    its wave-tile plan, and the exchange step."""

    def __init__(self, g, rank, world):
        from spla_amd import kernels
        self.g, self.rank, self.world = g, rank, world
        self.blk = g["blk"]
        self.n = g["n"]
        self.n_local = self.blk.n_local
        # the plan = wave-tile work items (built once per matrix on the device, like the AccCsr upload)
        self.plan = kernels.CsrPlan(self.n_local, g["nnz_local"], g["Ap"])
        self.max_rows = self.blk.max_rows
        self.n_cols_padded = self.blk.n_cols_padded

    def gather(self, v_local_padded, v_full):
        """all-gather(v) of the dense input vector: every rank contributes its (padded) slice."""
        return self.blk.all_gather(v_local_padded, v_full)


def bfs_state(sh: Sharded, Ax):
    """Run a BFS (pull steps through the HIP kernel) from the highest-degree vertex and return the depth vector
    and the peak-frontier level: the mid-BFS state the timed step replays.  Not timed."""
    from spla_amd import kernels
    g, world = sh.g, sh.world
    n_local, lo = sh.n_local, g["lo"]
    deg = (g["Ap"][1:] - g["Ap"][:-1]).to(torch.int64)
    best = torch.stack([deg.max(), deg.argmax() + lo]).clone().to(coll_device())
    if world > 1:
        allb = [torch.empty_like(best) for _ in range(world)]
        dist.all_gather(allb, best)
        best = max(allb, key=lambda x: (int(x[0]), -int(x[1])))
    src = int(best[1])
    depth = torch.zeros(n_local, dtype=torch.int32, device="cuda")
    front = torch.zeros(sh.max_rows if world > 1 else n_local, dtype=torch.int32, device="cuda")
    if lo <= src < g["hi"]:
        depth[src - lo] = 1
        front[src - lo] = 1
    v_full = torch.zeros(sh.n_cols_padded, dtype=torch.int32, device="cuda")
    new = torch.empty(n_local, dtype=torch.int32, device="cuda")
    sizes = []
    level = 1
    while True:
        sh.gather(front, v_full) if world > 1 else v_full.copy_(front)
        kernels.mxv_masked(sh.plan, "INT", sh.n_cols_padded, g["Ap"], g["Aj"], Ax, depth, v_full, new,
                           "BAND", "BOR", "EQZERO", 0, early_exit=False)
        level += 1
        hit = new != 0
        cnt = hit.sum().to(coll_device())
        if world > 1:
            dist.all_reduce(cnt)
        cnt = int(cnt)
        if cnt == 0:
            break
        depth[hit] = level
        front.zero_()
        front[:n_local] = hit.to(torch.int32)
        sizes.append(cnt)
    # sizes[k] = vertices discovered by the step that consumed the frontier {depth == k + 1}.  The timed step is the
    # one that PRODUCES the largest frontier (the heaviest pull step of the traversal): it consumes {depth == L}
    # with the rows visited up to level L masked out.
    L = 1 + int(np.argmax(sizes))
    return depth, L, sizes, src


def copy_ceiling_gbs(n_words=1 << 28, reps=10):
    """GB/s (read + write) of a 1 GiB device-to-device copy on the current stream: the streaming ceiling measured in the same run."""
    a = torch.ones(n_words, dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    del a, b
    return 2 * 4 * n_words / (float(np.median(ms)) * 1e-3) / 1e9


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    backend = os.environ.get("SPLA_DIST_BACKEND", "nccl")  # "gloo": debug mode, ranks may share one GPU
    device_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from spla_amd import _lib, kernels
    _lib.load()
    _lib.call("splahip_init", device_index)

    if args.workload == "sssp-step" and world > 1:  # BASELINE configs[3] as specified: strong 1-D row split of R-MAT scale 24 over the C-ABI dist layer
        import bench_dist
        line = bench_dist.sssp_step_sharded(args, rank, world)
        if rank == 0:
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.workload in ("tc-mxmT", "mxm") and world > 1:  # BASELINE configs[4] by rows of the mask / of A, B whole on every rank
        import bench_dist
        line = bench_dist.spgemm_sharded(args, rank, world)
        if rank == 0:
            print(json.dumps(line), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.workload == "spmv-fp32":
        import bench_spmv
        line = bench_spmv.run(args, rank, world)
        if world == 1 and not args.no_variants:  # the contract line of rounds 1-2 (BASELINE configs[1]) rides along as variants[0]
            import copy
            b = copy.copy(args)
            b.scale, b.no_scale23, b.steps, b.warmup = 22, True, max(20, args.steps), args.warmup
            bl = bfs_step_line(b, rank, world)
            keep = ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "verified", "cpu_baseline")
            v0 = {"name": "BFS step (BASELINE configs[1]; the contract line of rounds 1-2)", **{k: bl[k] for k in keep if k in bl}}
            v0["variants"] = [x for x in bl.get("variants", []) if x["name"].startswith("bfs-step")]
            line["variants"] = [v0] + line.pop("variants", []) + line.pop("variants_masked", []) + line.pop("variants_same_matrix", [])
        elif "variants_same_matrix" in line:
            line["variants"] = line.pop("variants", []) + line.pop("variants_masked", []) + line.pop("variants_same_matrix")
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload != "bfs-step":
        if world != 1:
            raise SystemExit("this operator workload runs on one GPU (sssp-step, tc-mxmT, mxm, bfs-step and the default spmv-fp32 take --gpus N)")
        import bench_workloads
        print(json.dumps(bench_workloads.WORKLOADS[args.workload](args)), flush=True)
        return
    out = bfs_step_line(args, rank, world)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bfs_step_line(args, rank, world):
    """BASELINE configs[1]: the BFS-step masked SpMV on R-MAT scale 22 (+ log2 N), the bench line of rounds 1-2."""
    from spla_amd import kernels
    g = build_graph(args, rank, world)
    sh = Sharded(g, rank, world)
    n_local, nnz_local = sh.n_local, g["nnz_local"]
    ones_i = torch.ones(nnz_local, dtype=torch.int32, device="cuda")

    depth, L, sizes, src = bfs_state(sh, ones_i)
    sh.plan_csr = sh.plan
    if args.plan == "sell":  # the timed step runs on the sliced-ELL copy (part of the resident matrix, like the plan's work items)
        t0 = time.perf_counter()
        sh.plan = kernels.CsrPlan(n_local, nnz_local, g["Ap"], sh.n_cols_padded, g["Aj"], ones_i)
        torch.cuda.synchronize()
        if rank == 0:
            log(f"[bench] sliced-ELL copy attached in {(time.perf_counter() - t0) * 1e3:.0f} ms: {sh.plan.sell}")
    mask = torch.where(depth <= L, depth, torch.zeros_like(depth)).contiguous()      # visited up to level L
    front = torch.zeros(sh.max_rows if world > 1 else n_local, dtype=torch.int32, device="cuda")
    front[:n_local] = (depth == L).to(torch.int32)
    deg = (g["Ap"][1:] - g["Ap"][:-1]).to(torch.int64)
    sel_rows = mask == 0
    nnz_sel_local = int(deg[sel_rows].sum())
    stat = torch.tensor([nnz_sel_local, int(sel_rows.sum()), int((depth == L).sum())], device=coll_device(), dtype=torch.int64)
    if world > 1:
        dist.all_reduce(stat)
    nnz_sel_total, rows_sel_total, front_total = (int(x) for x in stat.tolist())
    if rank == 0:
        log(f"[bench] BFS from {src}: frontier sizes per level {sizes}; timed level L={L}: frontier={front_total} "
            f"rows selected={rows_sel_total} nnz_sel={nnz_sel_total} ({100.0 * nnz_sel_total / g['nnz_global']:.1f}% of nnz)")

    v_full = torch.zeros(sh.n_cols_padded, dtype=torch.int32, device="cuda")
    r = torch.empty(n_local, dtype=torch.int32, device="cuda")
    if world == 1:
        v_full.copy_(front)

    # the exchange step (N > 1): dense all-gather of the padded slices, or — when the frontier is sparse — the sparsity-aware form behind the C ABI
    # (splahip_dist_exchange_sparse, dist.hip: positions only, bounded messages, no count exchange; sharding.SparseExchangeABI — over RCCL, or with a
    # host-staged transport between its pack / scatter kernels under the gloo debug backend).  SPLA_EXCHANGE=dense|sparse|torch-sparse|abi overrides the rule
    # (torch-sparse: round 1's chain of torch ops); the rule depends on global quantities only, so every rank picks the same form.
    mode = os.environ.get("SPLA_EXCHANGE", "auto")
    sparse = world > 1 and (mode in ("sparse", "torch-sparse") or (mode == "auto" and front_total * 16 < g["n"]))
    sx = None
    sx_comm = None
    if sparse:
        from spla_amd import sharding
        if mode == "torch-sparse":
            sx = sharding.SparseExchange(sh.blk, torch.int32, "cuda", fill=0)
        else:
            if dist.get_backend() == "nccl":
                try:
                    sx_comm = sharding.DistComm(world, rank)
                except Exception as exc:  # the host-staged transport stays available
                    log(f"[bench] splahip_dist_comm_create failed ({exc}); sparse exchange staged through torch.distributed")
            sx = sharding.SparseExchangeABI(sh.blk, torch.int32, "cuda", fill=0, comm=sx_comm)
        # the bound every rank pads its position list to: the largest per-rank frontier count — in a real traversal the previous level's
        # count reduction delivers it (a MAX beside the SUM); padding to the global count would send world x the bytes
        fmax = torch.tensor([int((depth == L).sum())], device=coll_device(), dtype=torch.int64)
        dist.all_reduce(fmax, op=dist.ReduceOp.MAX)
        front_bound = int(fmax[0])
        v_full = sx.exchange(front[:n_local], value=1, upper_bound=front_bound)[:sh.n_cols_padded]

    def exchange():
        if sparse:
            sx.exchange(front[:n_local], value=1, upper_bound=front_bound)
        else:
            sh.gather(front, v_full)

    # SPLA_EXCHANGE=abi: the whole step through the C-ABI dist layer (spla_amd/csrc/kernels/dist.hip): RCCL all-gather of the dense slice on the
    # communicator's stream, overlapped with the local-column part of the SpMV over the column-ordered blocks; the remote-column part folds onto it
    abi = None
    if world > 1 and mode == "abi":
        import bench_dist
        blk_x = sh.blk
        blk_x.Ax = ones_i
        abi = bench_dist.ShardedProduct(blk_x, ones_i, "INT", "BAND", "BOR", "EQZERO", 0, 0, world, rank, direct=os.environ.get("SPLA_ALLGATHER", "nccl") == "direct")
        v_full = abi.v_full

    def step(early=False):
        if abi is not None and not early:
            abi.step(mask, front, r)
            return
        if world > 1:
            exchange()
        kernels.mxv_masked(sh.plan, "INT", sh.n_cols_padded, g["Ap"], g["Aj"], ones_i, mask, v_full, r,
                           "BAND", "BOR", "EQZERO", 0, early_exit=early)

    def timed(fn, steps, warmup):
        """Contract timing: W untimed steps, then exactly K steps bracketed by barrier + synchronize, max over ranks;
        per-launch HIP events on the launch stream give the kernel's average duration."""
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
        per = [a.elapsed_time(b) for a, b in evs]  # ms
        if world > 1:
            w = torch.tensor([wall], device=coll_device(), dtype=torch.float64)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            wall = float(w[0])
        return wall, per

    wall, per = timed(step, args.steps, args.warmup)
    ms_per_step = wall / args.steps * 1e3
    gteps = nnz_sel_total / (wall / args.steps) / 1e9

    # roofline of the dominant kernel (this rank's launch): algorithmic bytes per launch / average launch duration.
    # B_mxv = 8*nnz_sel + 4*(n_rows+1) + 4*n_rows (mask) + 4*n_rows (r) + 4*n_cols (v once)   
    if world > 1:
        # kernel-only duration: time the local SpMV without the exchange
        _, per_k = timed(lambda: kernels.mxv_masked(sh.plan, "INT", sh.n_cols_padded, g["Ap"], g["Aj"], ones_i, mask,
                                                    v_full, r, "BAND", "BOR", "EQZERO", 0), args.steps, 2)
    else:
        per_k = per
    k_ms = float(np.mean(per_k))
    # a matrix whose stored values are all one word (the BFS step's adjacency matrix of ones) is served structure-only by the sliced-ELL
    # bit form: 4 B per entry (the column word), and the byte model says so ("report against 4*nnz_sel instead")
    structure_only = bool(args.plan == "sell" and sh.plan.sell and sh.plan.sell.get("uniform_values"))
    entry_bytes = 4 if structure_only else 8
    alg_bytes = entry_bytes * nnz_sel_local + 4 * (n_local + 1) + 8 * n_local + 4 * sh.n_cols_padded
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": kernel_name(sh.plan, "INT", "BAND", "BOR", False), "kernel_avg_us": round(k_ms * 1e3, 2),
                "kernel_median_us": round(float(np.median(per_k)) * 1e3, 2), "algorithmic_bytes": alg_bytes,
                "bytes_per_entry": entry_bytes,
                "byte_model": ("structure-only: every stored value is the same word, the kernel streams column words only; " if structure_only else "")
                              + "B = %d*nnz_sel + 4*(n_rows+1) + 4*n_rows (mask) + 4*n_rows (r) + 4*n_cols (v once)" % entry_bytes}
    if world == 1 and "first_hit" in roofline["kernel"]:
        fh = sh.plan.first_hit()
        roofline["byte_model"] += ("; the first-hit kernel (frontier vector over a structure-only matrix: every product is 0 or one word, BOR is idempotent) stops a row at its first "
                                   "hit and so reads fewer than the selected entries; `achieved` still divides the operation's algorithmic bytes by the call's time "
                                   "(pass over v + first-hit kernel)")
        roofline["first_hit"] = fh
        #  the kernel stops a row at its first hit: its primary figure is what it HAS to read — the entries up to and including each selected row's
        # first frontier column, 4 bytes each (column words only) — over the call's time; the figure on the operation's full algorithmic bytes (which a
        # kernel that reads less can push past what the HBM delivers) stays beside it under its own name
        req = required_entries_early_exit(g, mask, v_full)
        req_bytes = 4 * req + 4 * (n_local + 1) + 8 * n_local + sh.n_cols_padded // 8 + 4 * sh.n_cols_padded
        roofline["on_operation_bytes"] = {"achieved": roofline["achieved"], "frac": roofline["frac"], "algorithmic_bytes": alg_bytes}
        achieved = req_bytes / (k_ms * 1e-3) / 1e9
        roofline["achieved"], roofline["frac"] = round(achieved, 1), round(achieved / HBM_PEAK_GBS, 4)
        roofline["entries_required"], roofline["required_bytes"] = req, req_bytes
        roofline["basis"] = "required bytes of a first-hit walk: 4*entries_required + Ap + mask + r + one pass over v and its bitmap"

    # measured streaming ceiling of this GPU in this run: a 1 GiB device-to-device copy, read + write bytes / time.
    # `frac` stays against the nominal 8 TB/s; `frac_of_measured_copy` is the same numerator against what a plain copy reaches here.
    from bench_spmv import read_stream_gbs
    rs_gbs = read_stream_gbs()
    roofline["measured_read_stream_gbs"] = round(rs_gbs, 1)
    roofline["frac_of_measured_read_stream"] = round(achieved / rs_gbs, 4)
    copy_gbs = copy_ceiling_gbs()
    roofline["measured_copy_gbs"] = round(copy_gbs, 1)
    roofline["frac_of_measured_copy"] = round(achieved / copy_gbs, 4)

    # HBM-side traffic of the same kernel from the rocprofv3 PMC passes (tools/gpu_round.sh -> tools/pmc_traffic.py):
    # collected by a separate profiler run of this command, so it is read from profiles/ when it matches this workload
    tf = ROOT / "profiles" / "r06_pmc_traffic.json"
    if not tf.exists():
        tf = ROOT / "profiles" / "r05_pmc_traffic.json"
    if not tf.exists():
        tf = ROOT / "profiles" / "r04_pmc_traffic.json"
    if not tf.exists():
        tf = ROOT / "profiles" / "pmc_traffic.json"  # (round 2's pass)
    if world == 1 and tf.exists():
        t = json.loads(tf.read_text())
        if t.get("config", {}).get("nnz_selected") == nnz_sel_total and t.get("config", {}).get("nnz") == g["nnz_global"]:
            roofline["traffic"] = t["traffic_bytes"]
            roofline["traffic_source"] = f"profiles/{tf.name}: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate rocprofv3 --pmc passes"

    out = {
        "metric": "masked SpMV GTEPS", "value": round(gteps, 3), "unit": "GTEPS", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "exec_mxv_masked BFS step: R-MAT scale %d (ef %d, %s labels), BAND/BOR int32, mask=depth EQZERO at the level producing the largest frontier, early_exit=false"
                               % (g["gscale"], args.edge_factor, {False: "raw", True: "randomly permuted", "owner-sorted": "owner-sorted"}[g["permute"]]),
                   "n": g["n"], "nnz": g["nnz_global"], "nnz_selected": nnz_sel_total, "rows_selected": rows_sel_total,
                   "frontier": front_total, "bfs_level": L, "parallelism": f"1-D nnz-balanced row blocks x{world}" + (" + RCCL all-gather(v)" if world > 1 else ""),
                   "work_items": sh.plan.n_items, "long_rows": sh.plan.n_long},
        "roofline": roofline,
    }

    # ---- the exchange step on its own (SURVEY.md 8e reporting): all-gather time, per-link rate, overlap -----------
    if world > 1:
        _, per_x = timed(exchange, args.steps, 2)
        x_ms = float(np.mean(per_x))
        # what this rank receives from EACH peer (direct all-gather over the xGMI mesh): a padded dense slice, or the longest position list
        slice_bytes = 4 * sh.max_rows if not sparse else (4 * int(sx.prev.numel() // world) if mode == "torch-sparse" else 4 * int(sx.last_words))
        xm = torch.tensor([x_ms], device=coll_device(), dtype=torch.float64)
        dist.all_reduce(xm, op=dist.ReduceOp.MAX)
        if abi is not None:
            out["exchange_abi"] = bench_dist.phase_report(abi, mask, front, r, max(5, args.steps // 2), world)
        out["exchange"] = {"mode": "C ABI: splahip_dist_mxv_masked (RCCL all-gather on its own stream || local-column part)" if abi is not None else
                           ("sparse, torch ops (positions of the non-fill entries)" if mode == "torch-sparse" else
                            "C ABI: splahip_dist_exchange_sparse (pack kernel, RCCL all-gather of bounded position messages, scatter kernel)" if sx_comm is not None else
                            "C ABI: splahip_dist_sx_pack / _scatter kernels around a host-staged torch.distributed all-gather (debug backend)") if sparse else "dense (padded slices)",
                           "collective": "all_gather_into_tensor (%s)" % ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " debug backend"), "avg_us": round(x_ms * 1e3, 2), "max_over_ranks_us": round(float(xm[0]) * 1e3, 2),
                           "bytes_per_peer": slice_bytes, "bytes_received_per_rank": slice_bytes * (world - 1),
                           "gbs_per_link_if_direct": round(slice_bytes / (x_ms * 1e-3) / 1e9, 2),
                           "local_spmv_us": round(k_ms * 1e3, 2), "overlap_fraction": out["exchange_abi"]["overlap_fraction"] if abi is not None else 0.0,
                           "entries_exchanged": int(front_total) if sparse else int(g["n"]),
                           "note": ("exchange on the communicator's stream beside the local-column part (exchange_abi: the phases on their own)" if abi is not None else
                                    "torch.distributed exchange, then the SpMV, back to back on one stream (SPLA_EXCHANGE=abi: the overlapped C-ABI step; "
                                    "--workload sssp-step --gpus N uses it by default)")}

    # ---- verification of the timed result against the oracle (full local rows), and the CPU baseline ------------
    if rank == 0 and not args.no_verify:
        out["verified"] = verify_full(g, sh, mask, v_full, r, ones_i)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # the CPU baseline is reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args, g, sh, mask, v_full, r)

    if not args.no_variants:
        out["variants"] = variants(args, g, sh, mask, v_full, front, timed, world, rank, nnz_sel_total, nnz_sel_local)
    for owned in (sx, sx_comm, abi):  # device buffers / the RCCL communicator behind the C ABI
        if owned is not None and hasattr(owned, "close"):
            owned.close()
    return out


def verify_full(g, sh, mask, v_full, r, Ax):
    """HIP result of the timed step == oracle (plain-C restatement) on every local row, bit-exact (int32)."""
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    want = orc.mxv_masked("INT", sh.n_local, sh.n_cols_padded, g["Ap"].cpu().numpy(), g["Aj"].cpu().numpy(),
                          Ax.cpu().numpy(), mask.cpu().numpy(), v_full.cpu().numpy(), "BAND", "BOR", "EQZERO", 0)
    ok = bool(np.array_equal(want, r.cpu().numpy()))
    log(f"[bench] full-size parity vs oracle: {'OK' if ok else 'MISMATCH'} ({sh.n_local} rows)")
    if not ok:
        raise SystemExit("bench: HIP result differs from the oracle")
    return True


def cpu_baseline(args, g, sh, mask, v_full, r):
    """The reference CPU path timed on this host: the genuine reference build (oracle/_ref) when present, else the
    plain-C port; on a bounded sample of the same workload = every s-th row of rank 0's row block (s chosen so the
    sample holds ~cpu_sample_nnz entries; rows are independent, so the sampled rows' results are checked too)."""
    from oracle import pyoracle
    stride = max(1, int(round(g["nnz_local"] / max(1, args.cpu_sample_nnz))))
    rows_idx = torch.arange(0, sh.n_local, stride, device="cuda")
    Ap64 = g["Ap"].to(torch.int64)
    deg = (Ap64[1:] - Ap64[:-1])[rows_idx]
    Ap_s = torch.zeros(rows_idx.numel() + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(deg, 0, out=Ap_s[1:])
    nnz_s = int(Ap_s[-1])
    # entry k of sampled row i sits at Ap[rows_idx[i]] + (k - Ap_s[i])
    rep = torch.repeat_interleave(torch.arange(rows_idx.numel(), device="cuda"), deg)
    src = Ap64[rows_idx][rep] + (torch.arange(nnz_s, device="cuda") - Ap_s[:-1][rep])
    Aj_s = g["Aj"][src].cpu().numpy()
    Ax_s = np.ones(nnz_s, np.int32)
    rows = int(rows_idx.numel())
    Ap_s = Ap_s.to(torch.int32).cpu().numpy().view(np.uint32)
    mask_s = mask[rows_idx].cpu().numpy()
    v = v_full.cpu().numpy()
    nnz_sel = int(deg[mask[rows_idx] == 0].sum())
    impl = pyoracle.Reference() if pyoracle.have_reference() else pyoracle.Oracle()
    times, first = [], None
    if impl.kind == "reference":
        # keep the reference objects alive across calls so the timed region is the steady-state exec call only
        t = 0
        M = impl._matrix(t, rows, sh.n_cols_padded, Ap_s, Aj_s, Ax_s)
        hm, hv, hr = impl._dense_vector(t, mask_s), impl._dense_vector(t, v), impl._empty_vector(t, rows)
        hi = impl._scalar(0, t)
        opm, opa, ops = impl._op("Binary", "BAND", t), impl._op("Binary", "BOR", t), impl._op("Select", "EQZERO", t)
        for it in range(4):
            t0 = time.perf_counter()
            st = impl.lib.spla_Exec_mxv_masked(hr, hm, M, hv, opm, opa, ops, hi, None, None)
            dt = time.perf_counter() - t0
            assert st == 0
            if it > 0:
                times.append(dt)
            else:
                first = dt
        got = np.empty(rows, np.uint32)
        impl.lib.refshim_vector_dense_read(hr, got.ctypes.data_as(pyoracle._u32p))
        got = got.view(np.int32)
        impl._unref(M, hm, hv, hr, hi)
    else:
        for it in range(4):
            t0 = time.perf_counter()
            got = impl.mxv_masked("INT", rows, sh.n_cols_padded, Ap_s, Aj_s, Ax_s, mask_s, v, "BAND", "BOR", "EQZERO", 0)
            if it > 0:
                times.append(time.perf_counter() - t0)
    same = bool(np.array_equal(got, r[rows_idx].cpu().numpy()))
    med = float(np.median(times))
    log(f"[bench] cpu baseline ({impl.kind}): every {stride}th row: {rows} rows / {nnz_s} nnz / {nnz_sel} selected nnz, "
        f"median {med * 1e3:.1f} ms, HIP == {impl.kind}: {same}")
    if not same:
        raise SystemExit("bench: HIP result differs from the CPU reference on the sample")
    return {"value": round(nnz_sel / med / 1e9, 4), "unit": "GTEPS", "cores": 1, "kind": impl.kind,
            "sample": f"every {stride}th row of rank 0's row block: {rows} rows, {nnz_s} nnz, {nnz_sel} in selected rows, same "
                      f"mask/v; median of 3 steady-state exec_mxv_masked calls after 1 warm-up"
                      + (f" (first call incl. COO->LIL conversion {first * 1e3:.0f} ms)" if first else ""),
            "host_cpus": os.cpu_count(), "ms": round(med * 1e3, 2), "matches_hip": same}


def required_entries_early_exit(g, mask, v_full):
    """entries an early-exit BFS step has to read: per selected row, up to and including its first entry whose column is in the frontier"""
    Ap64 = g["Ap"].to(torch.int64)
    deg = Ap64[1:] - Ap64[:-1]
    n_local = deg.numel()
    rows = torch.repeat_interleave(torch.arange(n_local, device="cuda"), deg)
    hit = v_full[g["Aj"].to(torch.int64)] != 0
    pos = torch.arange(rows.numel(), device="cuda") - Ap64[:-1][rows]
    first = torch.full((n_local,), 1 << 40, dtype=torch.int64, device="cuda")
    first.scatter_reduce_(0, rows[hit], pos[hit], reduce="amin", include_self=True)
    need = torch.minimum(first + 1, deg)  # no hit: the whole row is read
    return int(need[mask == 0].sum())


def variants(args, g, sh, mask, v_full, front, timed, world, rank, nnz_sel_total, nnz_sel_local):
    """Other measured forms of the same kernel family on the same matrix (kernel-only, inputs resident)."""
    from spla_amd import _lib, kernels
    res = []
    n_local, nnz_local = sh.n_local, g["nnz_local"]
    steps = max(10, args.steps // 2)

    def run(name, t, Ax, m, v, r, mul, add, sel, init, early, nnz_sel, plan=None, entry_bytes=8):
        plan = plan or sh.plan_csr
        fn = lambda: kernels.mxv_masked(plan, t, sh.n_cols_padded, g["Ap"], g["Aj"], Ax, m, v, r, mul, add, sel, init, early_exit=early)
        _, per = timed(fn, steps, 3)
        ms = float(np.mean(per))
        b = entry_bytes * nnz_sel + 4 * (n_local + 1) + 8 * n_local + 4 * sh.n_cols_padded
        tot = torch.tensor([nnz_sel], device=coll_device(), dtype=torch.int64)
        if world > 1:
            dist.all_reduce(tot)
        res.append({"name": name, "kernel": kernels.mxv_variant(t, mul, add, early, plan), "kernel_avg_us": round(ms * 1e3, 2),
                    "gteps_local": round(nnz_sel / (ms * 1e-3) / 1e9, 2), "hbm_gbs": round(b / (ms * 1e-3) / 1e9, 1),
                    "roofline_frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "nnz_selected_local": nnz_sel, "bytes_per_entry": entry_bytes})

    r_i = torch.empty(n_local, dtype=torch.int32, device="cuda")
    ones_i = torch.ones(nnz_local, dtype=torch.int32, device="cuda")
    run("bfs-step early_exit=true (as reference bfs())", "INT", ones_i, mask, v_full, r_i, "BAND", "BOR", "EQZERO", 0, True, nnz_sel_local)
    # an early-exit variant reports against the bytes of the entries actually REQUIRED — up to and including the first entry of every
    # selected row whose product makes the sum differ from init (cpu_mxv.hpp:96-98: here the first entry whose column is in the frontier) — and
    # against the full nnz_sel.  The kernel's time divided into the full figure says nothing about bandwidth: it is listed, not claimed.
    req = required_entries_early_exit(g, mask, v_full)
    b_req = 8 * req + 4 * (n_local + 1) + 8 * n_local + 4 * sh.n_cols_padded
    ms_e = res[-1]["kernel_avg_us"] * 1e-3
    res[-1]["entries_required"] = req
    res[-1]["hbm_gbs_required_bytes"] = round(b_req / (ms_e * 1e-3) / 1e9, 1)
    res[-1]["roofline_frac_required_bytes"] = round(b_req / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    res[-1]["roofline_frac_full_nnz_sel"] = res[-1].pop("roofline_frac")
    res[-1]["note"] = "one lane per row walks its row until the first hit: the required-bytes fraction is the bandwidth figure; the full-nnz_sel fraction is listed as SURVEY 8d asks and is not a bandwidth claim"
    # the same early-exit call as the operator API's bfs() issues it: the matrix's owner has told the plan that every stored value is one word (spla_hip.h:
    # splahip_csr_plan_set_uniform_values; algos_hip.cpp does at upload), so the call runs on the first-hit kernel (mxv_first_hit.hpp) without any hint
    plan_u = kernels.CsrPlan(n_local, nnz_local, g["Ap"])
    plan_u.set_uniform_values(True, 1)
    run("bfs-step early_exit=true, the plan told that the matrix is structure-only (what the operator API's bfs() runs)", "INT", ones_i, mask, v_full, r_i, "BAND", "BOR", "EQZERO", 0, True,
        nnz_sel_local, plan=plan_u)
    ms_u = res[-1]["kernel_avg_us"] * 1e-3
    res[-1]["entries_required"] = req
    res[-1]["roofline_frac_required_bytes"] = round((4 * req + 4 * (n_local + 1) + 8 * n_local + 4 * sh.n_cols_padded) / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    res[-1]["roofline_frac_full_nnz_sel"] = res[-1].pop("roofline_frac")
    res[-1]["note"] = "pass over v (bitmap, one-word test) + first-hit kernel; required bytes at 4 B per required entry (column ids only)"
    plan_u.close()
    zeros_i = torch.zeros(n_local, dtype=torch.int32, device="cuda")
    run("all rows selected, BAND/BOR int32", "INT", ones_i, zeros_i, v_full, r_i, "BAND", "BOR", "EQZERO", 0, False, nnz_local)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(args.seed + 100 + rank)
    Ax_f = torch.rand(nnz_local, generator=gen, device="cuda", dtype=torch.float32)
    v_f = torch.rand(sh.n_cols_padded, generator=gen, device="cuda", dtype=torch.float32)
    r_f = torch.empty(n_local, dtype=torch.float32, device="cuda")
    zeros_f = torch.zeros(n_local, dtype=torch.float32, device="cuda")
    run("all rows selected, MULT/PLUS fp32 (headline bandwidth form)", "FLOAT", Ax_f, zeros_f, v_f, r_f, "MULT", "PLUS", "ALWAYS", 0.0, False, nnz_local)
    w_i = torch.randint(1, 256, (nnz_local,), generator=gen, device="cuda", dtype=torch.int32)
    d_i = torch.full((sh.n_cols_padded,), 1 << 30, dtype=torch.int32, device="cuda")
    d_i[torch.randint(0, sh.n_cols_padded, (sh.n_cols_padded // 8,), generator=gen, device="cuda")] = 1000
    run("all rows selected, PLUS/MIN int32 (SSSP step)", "INT", w_i, zeros_i, d_i, r_i, "PLUS", "MIN", "ALWAYS", 1 << 30, False, nnz_local)
    if world > 1:
        return res
    # the opt-in sliced-ELL + LDS hot-column copy of the same matrix (splahip_csr_plan_attach_sell), for the record (N = 1)
    r_ref = r_f.clone()
    sell_f = kernels.CsrPlan(n_local, nnz_local, g["Ap"], sh.n_cols_padded, g["Aj"], Ax_f)
    run("all rows selected, MULT/PLUS fp32, sliced-ELL copy", "FLOAT", Ax_f, zeros_f, v_f, r_f, "MULT", "PLUS", "ALWAYS", 0.0, False, nnz_local, plan=sell_f)
    run("all rows selected, MULT/PLUS fp32", "FLOAT", Ax_f, zeros_f, v_f, r_ref, "MULT", "PLUS", "ALWAYS", 0.0, False, nnz_local)
    res.pop()  # (re-run of the CSR form only to have its result at hand for the comparison)
    rel = ((r_f - r_ref).abs() / r_ref.abs().clamp_min(1e-30)).max().item() if n_local else 0.0
    res[-1]["max_rel_vs_csr_kernel"] = rel
    res[-1]["plan"] = sell_f.sell
    sell_f.close()
    sell_w = kernels.CsrPlan(n_local, nnz_local, g["Ap"], sh.n_cols_padded, g["Aj"], w_i)
    r_w = torch.empty_like(r_i)
    run("all rows selected, PLUS/MIN int32 (SSSP step), sliced-ELL copy (1/8 of v finite: bit form)", "INT", w_i, zeros_i, d_i, r_w, "PLUS", "MIN", "ALWAYS", 1 << 30, False,
        nnz_local, plan=sell_w)
    kernels.mxv_masked(sh.plan_csr, "INT", sh.n_cols_padded, g["Ap"], g["Aj"], w_i, zeros_i, d_i, r_i, "PLUS", "MIN", "ALWAYS", 1 << 30)
    torch.cuda.synchronize()
    res[-1]["identical_to_csr_kernel"] = bool(torch.equal(r_w, r_i))
    sell_w.close()
    # the column-ordered-blocks copy (splahip_csr_plan_attach_cob): the north-star form, fp32 plus-times over all rows
    t0 = time.perf_counter()
    cob_f = kernels.CsrPlan(n_local, nnz_local, g["Ap"])
    cob_info = cob_f.attach_cob(sh.n_cols_padded, g["Ap"], g["Aj"], Ax_f)
    torch.cuda.synchronize()
    cob_ms = (time.perf_counter() - t0) * 1e3
    r_c = torch.empty_like(r_f)
    run("all rows selected, MULT/PLUS fp32, column-ordered blocks copy (north-star form)", "FLOAT", Ax_f, zeros_f, v_f, r_c, "MULT", "PLUS", "ALWAYS", 0.0, False, nnz_local, plan=cob_f)
    res[-1]["max_rel_vs_csr_kernel"] = ((r_c - r_ref).abs() / r_ref.abs().clamp_min(1e-30)).max().item() if n_local else 0.0
    res[-1]["plan"] = cob_info
    res[-1]["plan_build_ms"] = round(cob_ms, 1)
    cob_f.close()
    cob_w = kernels.CsrPlan(n_local, nnz_local, g["Ap"])
    cob_w.attach_cob(sh.n_cols_padded, g["Ap"], g["Aj"], w_i)
    r_cw = torch.empty_like(r_i)
    run("all rows selected, PLUS/MIN int32 (SSSP step), column-ordered blocks copy", "INT", w_i, zeros_i, d_i, r_cw, "PLUS", "MIN", "ALWAYS", 1 << 30, False, nnz_local, plan=cob_w)
    res[-1]["identical_to_csr_kernel"] = bool(torch.equal(r_cw, r_i))
    cob_w.close()
    cob_i = kernels.CsrPlan(n_local, nnz_local, g["Ap"])
    cob_i.attach_cob(sh.n_cols_padded, g["Ap"], g["Aj"], ones_i)
    r_cb = torch.empty_like(r_i)
    run("bfs-step early_exit=false, column-ordered blocks copy (sparse-vector bit form, structure-only: all stored values equal)", "INT", ones_i, mask, v_full, r_cb,
        "BAND", "BOR", "EQZERO", 0, False, nnz_sel_local, plan=cob_i, entry_bytes=4)
    kernels.mxv_masked(sh.plan_csr, "INT", sh.n_cols_padded, g["Ap"], g["Aj"], ones_i, mask, v_full, r_i, "BAND", "BOR", "EQZERO", 0)
    torch.cuda.synchronize()
    res[-1]["identical_to_csr_kernel"] = bool(torch.equal(r_cb, r_i))
    cob_i.close()
    sell_i = kernels.CsrPlan(n_local, nnz_local, g["Ap"], sh.n_cols_padded, g["Aj"], ones_i)
    r_chk = torch.empty_like(r_i)
    uni = bool(sell_i.sell and sell_i.sell.get("uniform_values"))
    run("bfs-step early_exit=false, sliced-ELL copy" + (" (structure-only: all stored values equal)" if uni else ""), "INT", ones_i, mask, v_full, r_chk,
        "BAND", "BOR", "EQZERO", 0, False, nnz_sel_local, plan=sell_i, entry_bytes=4 if uni else 8)
    r_chk3 = torch.empty_like(r_i)
    os.environ["SPLAHIP_SELL_UNIFORM"] = "0"
    _lib.env_refresh()
    run("bfs-step early_exit=false, sliced-ELL copy, value words streamed (SPLAHIP_SELL_UNIFORM=0)", "INT", ones_i, mask, v_full, r_chk3, "BAND", "BOR", "EQZERO", 0,
        False, nnz_sel_local, plan=sell_i)
    os.environ.pop("SPLAHIP_SELL_UNIFORM")
    _lib.env_refresh()
    r_chk2 = torch.empty_like(r_i)
    os.environ["SPLAHIP_SELL_BITS"] = "0"
    _lib.env_refresh()
    run("bfs-step early_exit=false, sliced-ELL copy, value form forced (SPLAHIP_SELL_BITS=0)", "INT", ones_i, mask, v_full, r_chk2, "BAND", "BOR", "EQZERO", 0, False, nnz_sel_local, plan=sell_i)
    os.environ["SPLAHIP_SELL_BITS"] = "1"
    _lib.env_refresh()
    run("bfs-step early_exit=false, CSR wave-tile kernel (no sliced-ELL copy)", "INT", ones_i, mask, v_full, r_i, "BAND", "BOR", "EQZERO", 0, False, nnz_sel_local)
    torch.cuda.synchronize()
    res[-4]["identical_to_csr_kernel"] = bool(torch.equal(r_chk, r_i))
    res[-3]["identical_to_csr_kernel"] = bool(torch.equal(r_chk3, r_i))
    res[-2]["identical_to_csr_kernel"] = bool(torch.equal(r_chk2, r_i))
    sell_i.close()
    res.append(api_path_bfs_step(args, g, sh, mask, v_full, timed, nnz_sel_local, res))
    if args.scale == 22 and not args.no_scale23:
        res.append(fp32_scale23(args, timed))
    return res


def api_path_bfs_step(args, g, sh, mask, v_full, timed, nnz_sel_local, res):
    """The bench step through the operator API: spla_Exec_mxv_masked on Matrix / Vector objects (include/spla.h), the matrix prepared with
    Matrix::set_format(AccCsr) = plan + layout copies built before the first product.  What an API user gets, next to the kernel-level number."""
    from spla_amd import api
    api.set_accelerator(api.ACCELERATOR_HIP)
    n, n_cols = sh.n_local, sh.n_cols_padded
    t0 = time.perf_counter()
    M = api.Matrix.from_csr(g["Ap"].cpu().numpy().view(np.uint32), g["Aj"].cpu().numpy().view(np.uint32), np.ones(g["nnz_local"], np.int32), (n, n_cols), api.INT)
    M.set_format(6)  # FormatMatrix::AccCsr: upload + plan + both layout copies
    hm = api.Vector.from_lists(np.arange(n, dtype=np.uint32), mask.cpu().numpy(), n, api.INT)
    hv = api.Vector.from_lists(np.arange(n_cols, dtype=np.uint32), v_full.cpu().numpy(), n_cols, api.INT)
    hr = api.Vector(n, api.INT)
    init = api.Scalar(api.INT, 0)
    opm, opa, ops = api.INT.BAND, api.INT.BOR, api.INT.EQZERO
    prep_s = time.perf_counter() - t0
    fn = lambda: M.mxv(hm, hv, opm, opa, ops, out=hr, init=init)
    fn()
    steps = max(10, args.steps // 2)
    _, per = timed(fn, steps, 3)
    ms = float(np.mean(per))
    ri, rx = hr.read()
    got = np.zeros(n, np.int32)
    got[ri] = rx
    torch.cuda.synchronize()
    ref = [x for x in res if x["name"].startswith("bfs-step early_exit=false, sliced-ELL copy")][0]
    same = None
    try:
        r_k = torch.empty(n, dtype=torch.int32, device="cuda")
        from spla_amd import kernels
        kernels.mxv_masked(sh.plan, "INT", n_cols, g["Ap"], g["Aj"], torch.ones(g["nnz_local"], dtype=torch.int32, device="cuda"), mask, v_full, r_k, "BAND", "BOR", "EQZERO", 0)
        same = bool(np.array_equal(got, r_k.cpu().numpy()))
    except Exception as exc:  # pragma: no cover
        log(f"[bench] api-path comparison failed: {exc}")
    return {"name": "bfs-step early_exit=false through spla_Exec_mxv_masked (operator API, matrix prepared with set_format(AccCsr))", "kernel": "operator API -> dispatcher -> splahip_mxv_masked",
            "kernel_avg_us": round(ms * 1e3, 2), "gteps_local": round(nnz_sel_local / (ms * 1e-3) / 1e9, 2), "kernel_level_us": ref["kernel_avg_us"],
            "ratio_to_kernel_level": round(ms * 1e3 / ref["kernel_avg_us"], 3), "identical_to_kernel_level": same,
            "host_prepare_s": round(prep_s, 2), "note": "time per call as HIP events around the API call (asynchronous, like the kernel-level call); host_prepare_s = COO build + COO->CSR + upload + plan + copies"}


def fp32_scale23(args, timed):
    """The north-star configuration: fp32 plus-times masked SpMV (all rows) on a ~100 M-nnz scale-free CSR = R-MAT scale 23 (131 M nnz)."""
    from spla_amd import graphgen, kernels
    n, Ap, Aj = graphgen.rmat_csr(23, args.edge_factor, seed=args.seed, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(args.seed + 300)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    zeros = torch.zeros(n, dtype=torch.float32, device="cuda")
    out = {}
    b = 8 * nnz + 4 * (n + 1) + 8 * n + 4 * n
    for name, attach in (("csr wave-tile", None), ("sliced-ELL copy", "sell"), ("column-ordered blocks copy", "cob")):
        plan = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax) if attach == "sell" else kernels.CsrPlan(n, nnz, Ap)
        if attach == "cob":
            plan.attach_cob(n, Ap, Aj, Ax)
        r = torch.empty(n, dtype=torch.float32, device="cuda")
        _, per = timed(lambda: kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, zeros, v, r, "MULT", "PLUS", "ALWAYS", 0.0), max(10, args.steps // 2), 3)
        ms = float(np.mean(per))
        out[name] = {"kernel": kernels.mxv_variant("FLOAT", "MULT", "PLUS", False, plan), "kernel_avg_us": round(ms * 1e3, 2), "gteps": round(nnz / (ms * 1e-3) / 1e9, 2),
                     "hbm_gbs": round(b / (ms * 1e-3) / 1e9, 1), "roofline_frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if attach == "cob":
            exact = torch.zeros(n, dtype=torch.float64, device="cuda")
            rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
            exact.index_add_(0, rows, Ax.to(torch.float64) * v[Aj.to(torch.int64)].to(torch.float64))
            out[name]["max_rel_vs_fp64"] = ((r.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
            del rows, exact
        plan.close()
    best = out["column-ordered blocks copy"]
    return {"name": "NORTH STAR: all rows, MULT/PLUS fp32, R-MAT scale 23 (%d nnz, the ~100 M-nnz scale-free configuration)" % nnz, "kernel": best["kernel"],
            "kernel_avg_us": best["kernel_avg_us"], "gteps_local": best["gteps"], "hbm_gbs": best["hbm_gbs"], "roofline_frac": best["roofline_frac"],
            "nnz_selected_local": nnz, "bytes_per_entry": 8, "algorithmic_bytes": b, "forms": out}


if __name__ == "__main__":
    main()
