"""The other hot-path operators as bench.py workloads (`python bench.py --workload <name>`): same JSON contract as the default
BFS-step line (metric / value / roofline / cpu_baseline), one operator each, inputs resident in HBM, results checked against
the oracle, the genuine reference CPU backend timed beside it on a bounded sample.

    pagerank-vxm : exec_vxm_masked plus-times fp32, every entry of v stored (BASELINE configs[2]; com-Orkut is not on the
                   box, stand-in: symmetrised R-MAT scale 22 edge factor 28, values 0.85/outdeg)  -> pull form over M^T
    sssp-step    : exec_mxv_masked min-plus int32 on R-MAT scale 24 (BASELINE configs[3] at N = 1)
    tc-mxmT      : exec_mxmT_masked(L, L, L) MULT/PLUS/GTZERO int32, L = strict lower triangle of symmetrised R-MAT scale 20
                   (BASELINE configs[4], triangle-count form)
    mxm          : exec_mxm plus-times fp32 A x A on R-MAT (BASELINE configs[4]; scale chosen so the expanded products fit)
    transpose    : exec_m_transpose of the R-MAT scale-22 adjacency matrix (the cached M^T a pull-form vxm wants), int32, AINV on the way
    algorithms   : bfs / sssp / pr / tc end to end through the operator API (spla_Algorithm_*, every step on the device), on a
                   symmetrised R-MAT; wall-clock per call including the per-iteration host synchronisations, like the reference's
                   own whole-algorithm charts
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def _log(*a):
    import sys
    print(*a, file=sys.stderr, flush=True)


def _time_calls(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, [a.elapsed_time(b) for a, b in evs]


def _line(metric, unit, value, args, ms, dtype, config, roofline, extra=None):
    out = {"metric": metric, "value": round(value, 3), "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
           "data": "synthetic", "config": config, "roofline": roofline}
    if extra:
        out.update(extra)
    return out


def _roof(alg_bytes, ms, kernel, pmc=None):
    """`pmc`: name of the committed counter pass of this workload (profiles/r03_pmc_<pmc>.json or r02_pmc_<pmc>.json, written by tools/gpu_pmc_r0*.sh in
    separate rocprofv3 --pmc runs of the same command): its per-call fabric traffic becomes `traffic` (a builder-side figure, like
    bench.py's pmc_traffic.json — counters need their own pass)."""
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": None, "kernel": kernel, "kernel_avg_us": round(ms * 1e3, 2), "algorithmic_bytes": int(alg_bytes)}
    if pmc:
        import json
        from pathlib import Path
        f = Path(__file__).resolve().parent / "profiles" / f"r06_pmc_{pmc}.json"
        if not f.exists():
            f = Path(__file__).resolve().parent / "profiles" / f"r05_pmc_{pmc}.json"
        if not f.exists():
            f = Path(__file__).resolve().parent / "profiles" / f"r04_pmc_{pmc}.json"
        if not f.exists():
            f = Path(__file__).resolve().parent / "profiles" / f"r03_pmc_{pmc}.json"
        if not f.exists():  # (a workload whose kernels did not change since round 2 keeps that round's pass; the file name is printed as traffic_source)
            f = Path(__file__).resolve().parent / "profiles" / f"r02_pmc_{pmc}.json"
        try:
            per_call = json.loads(f.read_text()).get("_per_call")
            if per_call:
                roof["traffic"] = per_call["traffic_bytes"]
                roof["traffic_over_algorithmic"] = round(per_call["traffic_bytes"] / alg_bytes, 3)
                roof["l1_to_l2_read_requests"] = per_call["tcp_tcc_read_req"]
                roof["traffic_source"] = f"profiles/{f.name}: (2*FETCH_SIZE + WRITE_SIZE)*1024 per call, separate rocprofv3 --pmc passes"
        except (OSError, ValueError, KeyError):
            pass
    return roof


def _sym_rmat(scale, edge_factor, seed):
    """Symmetrised R-MAT pattern (both directions, self loops and duplicates removed): (n, Ap, Aj) on the GPU."""
    from spla_amd import graphgen
    n = 1 << scale
    key = graphgen.rmat_edge_keys(scale, edge_factor, seed=seed, device="cuda")
    i, j = torch.div(key, n, rounding_mode="floor"), key % n
    key = torch.unique(torch.cat([key, j * n + i]), sorted=True)
    del i, j
    Ap, Aj = graphgen.keys_to_csr(key, n, n)
    return n, Ap, Aj


# ---------------------------------------------------------------------------------------------------------------- vxm
def _orkut_like(args):
    """BASELINE configs[2]'s matrix: SuiteSparse com-Orkut — 3,072,441 vertices, 234.4 M stored entries (117.2 M undirected edges), largest degree 33,313
    (reference README.md:148; loader src/io.cpp:159-169).  Read through the MtxLoader mirror when SPLA_COM_ORKUT names the file; otherwise a symmetric Chung-Lu
    graph matched to those three numbers (graphgen.chung_lu_csr).  `--scale S` other than the default shrinks n and nnz by 2^(22 - S) (tests, debug runs)."""
    from spla_amd import graphgen
    path = os.environ.get("SPLA_COM_ORKUT")
    if path and os.path.exists(path):
        from spla_amd import api
        n, _, hAi, hAj = api.load_mtx(path, offset_indices=True, make_undirected=True, remove_loops=True)  # (this library's mirror of the reference's MtxLoader)
        Ai, Aj = torch.from_numpy(np.asarray(hAi).astype(np.int64)).cuda(), torch.from_numpy(np.asarray(hAj).astype(np.int64)).cuda()
        n = int(n)
        key = torch.unique(Ai * n + Aj, sorted=True)
        Ap, Aj = graphgen.keys_to_csr(key, n, n)
        return n, Ap, Aj, f"SuiteSparse com-Orkut ({os.path.basename(path)})"
    shrink = 1 << max(0, 22 - args.scale) if args.scale < 22 else 1
    n, nnz_t, dmax = 3_072_441 // shrink, 234_400_000 // shrink, max(64, 33_313 // max(1, int(shrink ** 0.5)))
    n, Ap, Aj = graphgen.chung_lu_csr(n, nnz_t, dmax, seed=args.seed, device="cuda")
    return n, Ap, Aj, (f"symmetric Chung-Lu graph matched to com-Orkut's n = 3,072,441, 234.4 M entries, max degree 33,313 (random labels; the file is not on the box)"
                       if shrink == 1 else f"Chung-Lu graph, com-Orkut's numbers shrunk by {shrink}")


def pagerank_vxm(args):
    from oracle import pyoracle
    from spla_amd import kernels
    if os.environ.get("SPLA_BENCH_PR_GRAPH", "orkut-like") == "rmat":  # rounds 1-3's stand-in: symmetrised R-MAT scale 22, edge factor 28
        scale, ef = args.scale, 28
        n, Ap, Aj = _sym_rmat(scale, ef, args.seed)
        what = f"symmetrised R-MAT scale {scale} ef {ef} (rounds 1-3's stand-in for com-Orkut)"
    else:
        n, Ap, Aj, what = _orkut_like(args)
    nnz = int(Ap[-1])
    dmax_meas = int((Ap[1:] - Ap[:-1]).max())
    deg = (Ap[1:] - Ap[:-1]).to(torch.float32)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
    Ax = (0.85 / deg.clamp(min=1.0))[rows].contiguous()  # examples/pr.cpp:81-88: 0.85 / outdeg(row)
    del rows
    _log(f"[bench] pagerank-vxm: {what}: n={n} nnz={nnz} max degree={dmax_meas}")
    Tp, Tj, Tx = kernels.csr_transpose(n, n, Ap, Aj, Ax)
    mode = os.environ.get("SPLA_BENCH_PLAN", "cob")  # layout copy on the transposed matrix: column-ordered blocks (default), sliced-ELL, or the CSR kernels only
    use_sell = mode == "sell"
    plan_T = kernels.CsrPlan(n, nnz, Tp, n, Tj, Tx) if use_sell else kernels.CsrPlan(n, nnz, Tp)
    if mode == "cob":
        plan_T.attach_cob(n, Tp, Tj, Tx)
    vi = torch.arange(n, device="cuda", dtype=torch.int32)
    vx = torch.full((n,), 1.0 / n, device="cuda", dtype=torch.float32)
    mask = torch.zeros(n, device="cuda", dtype=torch.float32)  # the dummy mask of pr(): ALWAYS ignores it
    out = {}

    def step():
        out["r"] = kernels.vxm_pull(plan_T, "FLOAT", n, n, Tp, Tj, Tx, mask, vi, vx, "MULT", "PLUS", "ALWAYS", identity=True)

    wall, per = _time_calls(step, args.steps, args.warmup)
    ms = float(np.mean(per))
    ri, rx = out["r"]
    # B_vxm (dense frontier = mxv over M^T): CSR stream + row pointers + mask + v + (value, touched) + compaction
    alg = 8 * nnz + 4 * (n + 1) + 4 * n + 4 * n + 8 * n + 8 * n + 8 * int(ri.numel())
    line = _line("masked vxm GTEPS (PageRank step)", "GTEPS", nnz / (wall / args.steps) / 1e9, args, wall / args.steps * 1e3, "f32",
                 {"workload": f"exec_vxm_masked MULT/PLUS fp32, ALWAYS mask, every v entry stored (a PageRank step); {what}", "n": n, "nnz": nnz, "max_degree": dmax_meas,
                  "out_entries": int(ri.numel()),
                  "form": ("pull over the column-ordered-blocks copy of the transposed matrix (cob_vxm: fp64 LDS accumulation)" if mode == "cob" else
                           "pull over the sliced-ELL copy of the transposed matrix (sell_vxm: value form, vxm mode)" if use_sell else
                           "pull over the transposed CSR (spmv_wavetile, VXM mode)") + " + order-preserving compaction"},
                 _roof(alg, ms, ("cob_vxm" if mode == "cob" else "sell_vxm" if use_sell else "spmv_wavetile<VXM>") + " + compaction",
                       pmc="pagerank_vxm_orkut_like" if mode == "cob" and nnz == 234853298 else None))
    if not args.no_verify:
        # structure: identical to the oracle's (same touched columns, index-sorted).  values: the oracle (= the reference) folds
        # every column sequentially in fp32; columns here have up to ~1e5 products, where that sequential fold itself is ~2e-5 off
        # the exact sum while the wave-tile tree is ~5e-7 off.  So: (1) HIP within 1e-5 of the fp64 sum for every entry,
        # (2) HIP within 1e-5 of the oracle for every column of <= 4096 products, (3) within the sequential fold's own error
        # bound len * 2^-24 for the longer ones (measured: the oracle is up to 1.8e-4 off the fp64 sum on a 239 K-product column).
        orc = pyoracle.Oracle()
        wi, wx = orc.vxm_masked("FLOAT", n, n, Ap.cpu().numpy(), Aj.cpu().numpy(), Ax.cpu().numpy(), mask.cpu().numpy(),
                                vi.cpu().numpy(), vx.cpu().numpy(), "MULT", "PLUS", "ALWAYS")
        gi, gx = ri.cpu().numpy().view(np.uint32), rx.cpu().numpy()
        ok = bool(np.array_equal(wi, gi))
        if ok:
            exact = torch.zeros(n, dtype=torch.float64, device="cuda")
            exact.index_add_(0, Aj.to(torch.int64), Ax.to(torch.float64) * (1.0 / n))
            ex = exact.cpu().numpy()[gi]
            lens = np.diff(Tp.cpu().numpy().astype(np.int64))[gi]
            rel_exact = np.abs(gx - ex) / np.abs(ex)
            rel_orc = np.abs(gx - wx) / np.abs(wx)
            tol = np.where(lens <= 4096, 1e-5, np.maximum(1e-5, lens * 2.0 ** -24))  # (n-1)*u: the textbook bound of a sequential fp32 sum
            ok = bool(rel_exact.max() <= 1e-5 and np.all(rel_orc <= tol))
            line["verify"] = {"max_rel_vs_fp64": float(rel_exact.max()), "max_rel_vs_oracle": float(rel_orc.max()),
                              "oracle_max_rel_vs_fp64": float((np.abs(wx - ex) / np.abs(ex)).max()), "longest_column": int(lens.max())}
        _log(f"[bench] vxm full-size parity: structure == oracle, values {line.get('verify')}: {'OK' if ok else 'MISMATCH'}")
        if not ok:
            raise SystemExit("bench: vxm result differs from the oracle")
        line["verified"] = True
    if not args.no_cpu_baseline and pyoracle.have_reference():
        ref = pyoracle.Reference()
        stride = 16
        svi = np.arange(0, n, stride, dtype=np.uint32)
        svx = np.full(len(svi), 1.0 / n, np.float32)
        prod = int((Ap[1:] - Ap[:-1])[::stride].sum())
        Ap_h, Aj_h, Ax_h, mk = Ap.cpu().numpy(), Aj.cpu().numpy(), Ax.cpu().numpy(), mask.cpu().numpy()
        t0 = time.perf_counter()
        ref.vxm_masked("FLOAT", n, n, Ap_h, Aj_h, Ax_h, mk, svi, svx, "MULT", "PLUS", "ALWAYS")
        first = time.perf_counter() - t0
        # steady state: exec only (objects rebuilt per call in the driver, so time a second full call minus nothing: report the exec-dominated call)
        hM = ref._matrix(2, n, n, Ap_h, Aj_h, Ax_h)
        hm, hr, hv = ref._dense_vector(2, mk), ref._empty_vector(2, n), ref._empty_vector(2, n)
        k1, k2 = ref._view(svi), ref._view(svx)
        ref.lib.spla_Vector_build(hv, k1, k2)
        hi = ref._scalar(0.0, 2)
        ops = (ref._op("Binary", "MULT", 2), ref._op("Binary", "PLUS", 2), ref._op("Select", "ALWAYS", 2))
        times = []
        for it in range(4):
            t0 = time.perf_counter()
            st = ref.lib.spla_Exec_vxm_masked(hr, hm, hv, hM, ops[0], ops[1], ops[2], hi, None, None)
            if it:
                times.append(time.perf_counter() - t0)
            assert st == 0
        med = float(np.median(times))
        line["cpu_baseline"] = {"value": round(prod / med / 1e9, 4), "unit": "GTEPS", "cores": 1, "kind": "reference",
                                "sample": f"frontier = every {stride}th vertex ({len(svi)} entries, {prod} products) on the same matrix; median of 3 "
                                          f"steady-state exec_vxm_masked calls (first full call incl. conversions {first * 1e3:.0f} ms)",
                                "host_cpus": os.cpu_count(), "ms": round(med * 1e3, 2)}
        ref._unref(hM, hm, hr, hv, k1, k2, hi)
    return line


# --------------------------------------------------------------------------------------------------------------- sssp
def sssp_step(args):
    from oracle import pyoracle
    from spla_amd import graphgen, kernels
    scale = 24 if args.scale == 22 else args.scale
    n, Ap, Aj = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda")
    nnz = int(Ap[-1])
    g = torch.Generator(device="cuda")
    g.manual_seed(args.seed + 5)
    Ax = torch.randint(1, 256, (nnz,), generator=g, device="cuda", dtype=torch.int32)
    INF = 1 << 30
    v = torch.full((n,), INF, dtype=torch.int32, device="cuda")
    idx = torch.randint(0, n, (n // 8,), generator=g, device="cuda")
    v[idx] = torch.randint(0, 5000, (n // 8,), generator=g, device="cuda", dtype=torch.int32)
    mask = torch.zeros(n, dtype=torch.int32, device="cuda")
    r = torch.empty(n, dtype=torch.int32, device="cuda")
    mode = os.environ.get("SPLA_BENCH_PLAN", "cob")  # column-ordered blocks (default), the sliced-ELL copy, or the CSR kernels only
    plan = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax) if mode == "sell" else kernels.CsrPlan(n, nnz, Ap)
    if mode == "cob":  # default geometry (2^14-row ranges): the step runs the sparse-input-vector form of the min-plus family (1/8 of v finite)
        plan.attach_cob(n, Ap, Aj, Ax, row_bits=int(os.environ.get("SPLA_BENCH_COB_ROW_BITS", 0)))
    _log(f"[bench] sssp-step: R-MAT scale {scale}: n={n} nnz={nnz} items={plan.n_items} sliced-ELL copy: {plan.sell} column-ordered blocks: {plan.cob}")
    step = lambda: kernels.mxv_masked(plan, "INT", n, Ap, Aj, Ax, mask, v, r, "PLUS", "MIN", "ALWAYS", INF)
    wall, per = _time_calls(step, args.steps, args.warmup)
    ms = float(np.mean(per))
    alg = 8 * nnz + 4 * (n + 1) + 8 * n + 4 * n
    line = _line("masked SpMV GTEPS (SSSP step)", "GTEPS", nnz / (wall / args.steps) / 1e9, args, wall / args.steps * 1e3, "int32",
                 {"workload": f"exec_mxv_masked PLUS/MIN int32 (min-plus), ALWAYS, init 2^30, weights 1..255, 1/8 of v finite; R-MAT scale {scale} "
                              f"ef {args.edge_factor}, raw labels", "n": n, "nnz": nnz, "work_items": plan.n_items},
                 _roof(alg, ms, kernels.mxv_variant("INT", "PLUS", "MIN", False, plan), pmc="sssp_step" if mode == "cob" and scale == 24 else None))
    if not args.no_verify:
        orc = pyoracle.Oracle()
        want = orc.mxv_masked("INT", n, n, Ap.cpu().numpy(), Aj.cpu().numpy(), Ax.cpu().numpy(), mask.cpu().numpy(), v.cpu().numpy(), "PLUS", "MIN", "ALWAYS", INF)
        ok = bool(np.array_equal(want, r.cpu().numpy()))
        _log(f"[bench] sssp-step full-size parity vs oracle: {'OK' if ok else 'MISMATCH'}")
        if not ok:
            raise SystemExit("bench: HIP result differs from the oracle")
        line["verified"] = True
    if not args.no_cpu_baseline and pyoracle.have_reference():
        ref = pyoracle.Reference()
        stride = max(1, nnz // args.cpu_sample_nnz)
        rows = torch.arange(0, n, stride, device="cuda")
        Ap64 = Ap.to(torch.int64)
        deg = (Ap64[1:] - Ap64[:-1])[rows]
        Aps = torch.zeros(rows.numel() + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(deg, 0, out=Aps[1:])
        rep = torch.repeat_interleave(torch.arange(rows.numel(), device="cuda"), deg)
        src = Ap64[rows][rep] + (torch.arange(int(Aps[-1]), device="cuda") - Aps[:-1][rep])
        Ap_s, Aj_s, Ax_s = Aps.to(torch.int32).cpu().numpy(), Aj[src].cpu().numpy(), Ax[src].cpu().numpy()
        m = int(rows.numel())
        M = ref._matrix(0, m, n, Ap_s, Aj_s, Ax_s)
        hm, hv, hr = ref._dense_vector(0, np.zeros(m, np.int32)), ref._dense_vector(0, v.cpu().numpy()), ref._empty_vector(0, m)
        hi = ref._scalar(INF, 0)
        ops = (ref._op("Binary", "PLUS", 0), ref._op("Binary", "MIN", 0), ref._op("Select", "ALWAYS", 0))
        times = []
        for it in range(4):
            t0 = time.perf_counter()
            assert ref.lib.spla_Exec_mxv_masked(hr, hm, M, hv, ops[0], ops[1], ops[2], hi, None, None) == 0
            if it:
                times.append(time.perf_counter() - t0)
        got = np.empty(m, np.uint32)
        ref.lib.refshim_vector_dense_read(hr, got.ctypes.data_as(pyoracle._u32p))
        same = bool(np.array_equal(got.view(np.int32), r[rows].cpu().numpy()))
        med = float(np.median(times))
        line["cpu_baseline"] = {"value": round(int(Aps[-1]) / med / 1e9, 4), "unit": "GTEPS", "cores": 1, "kind": "reference",
                                "sample": f"every {stride}th row: {m} rows, {int(Aps[-1])} nnz, same v; median of 3 steady-state exec_mxv_masked calls",
                                "host_cpus": os.cpu_count(), "ms": round(med * 1e3, 2), "matches_hip": same}
        ref._unref(M, hm, hv, hr, hi)
        if not same:
            raise SystemExit("bench: HIP result differs from the CPU reference on the sample")
    return line


def _mxm_info():
    import ctypes as C
    from spla_amd import _lib
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _lib.call("splahip_mxm_last_info", C.byref(a), C.byref(b), C.byref(c))
    return {"heavy_items": a.value, "light_rows_nnz": b.value, "dense_corner_h": c.value}


# --------------------------------------------------------------------------------------------------------------- mxmT
def _row_sample_csr(Ap, Aj, Ax, rows):
    """CSR (on the host, numpy) that keeps only the given rows' entries, other rows empty (same shape)."""
    n = Ap.numel() - 1
    Ap64 = Ap.to(torch.int64)
    keep = torch.zeros(n, dtype=torch.bool, device=Ap.device)
    keep[rows] = True
    deg = torch.where(keep, Ap64[1:] - Ap64[:-1], torch.zeros_like(Ap64[1:]))
    Aps = torch.zeros(n + 1, dtype=torch.int64, device=Ap.device)
    torch.cumsum(deg, 0, out=Aps[1:])
    ent = torch.repeat_interleave(keep, (Ap64[1:] - Ap64[:-1]))
    return Aps.to(torch.int32), Aj[ent].contiguous(), Ax[ent].contiguous()


def tc_mxmT(args):
    from oracle import pyoracle
    from spla_amd import graphgen, kernels
    scale = 20 if args.scale == 22 else args.scale
    n, Ap, Aj = _sym_rmat(scale, args.edge_factor, args.seed)
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
    low = Aj.to(torch.int64) < rows  # strict lower triangle (examples/tc.cpp:82-86)
    key = rows[low] * n + Aj[low].to(torch.int64)
    Lp, Lj = graphgen.keys_to_csr(key, n, n)
    nnz = int(Lp[-1])
    Lx = torch.ones(nnz, dtype=torch.int32, device="cuda")
    L = (Lp, Lj, Lx)
    deg = (Lp[1:] - Lp[:-1]).to(torch.int64)
    lrows = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    work = int((deg[lrows] + deg[Lj.to(torch.int64)]).sum())
    _log(f"[bench] tc-mxmT: L = strict lower triangle of symmetrised R-MAT scale {scale}: n={n} nnz(L)={nnz} merge work={work}")
    out = {}

    def step():
        out["R"] = kernels.mxmT_masked("INT", n, L, L, L, "MULT", "PLUS", "GTZERO", 0)

    wall, per = _time_calls(step, args.steps, args.warmup)
    ms = float(np.mean(per))
    Rp, Rj, Rx = out["R"]
    nnz_r = int(Rj.numel())
    triangles = int(Rx.to(torch.int64).sum())
    alg = 8 * (3 * nnz + nnz_r) + 4 * 4 * (n + 1)  # mask + A + B + R (SURVEY 8d compulsory bytes)
    line = _line("masked SpGEMM (mxmT) G merge-steps/s", "Gsteps/s", work / (wall / args.steps) / 1e9, args, wall / args.steps * 1e3, "int32",
                 {"workload": f"exec_mxmT_masked(L, L, L) MULT/PLUS/GTZERO int32 (triangle-count form), L = strict lower triangle of symmetrised R-MAT "
                              f"scale {scale} ef {args.edge_factor}", "n": n, "nnz_L": nnz, "merge_steps": work, "nnz_R": nnz_r, "triangles": triangles},
                 _roof(alg, ms, "mxmT_entry + mxmT_heavy + compaction", pmc="tc_mxmT" if scale == 20 and not os.environ.get("SPLAHIP_MXMT_ROW") else None))
    if not args.no_verify or not args.no_cpu_baseline:
        # restrict the mask to every 64th row: the same call on the sample is checked against the oracle and timed on the reference
        srows = torch.arange(0, n, 64, device="cuda")
        Mp_s, Mj_s, Mx_s = _row_sample_csr(Lp, Lj, Lx, srows)
        got = kernels.mxmT_masked("INT", n, (Mp_s, Mj_s, Mx_s), L, L, "MULT", "PLUS", "GTZERO", 0)
        Lh = tuple(t.cpu().numpy() for t in L)
        Mh = (Mp_s.cpu().numpy(), Mj_s.cpu().numpy(), Mx_s.cpu().numpy())
        if not args.no_verify:
            want = pyoracle.Oracle().mxmT_masked("INT", n, Mh, Lh, Lh, "MULT", "PLUS", "GTZERO", 0)
            ok = all(np.array_equal(np.asarray(a).view(np.uint32), b.cpu().numpy().view(np.uint32)) for a, b in zip(want, got))
            _log(f"[bench] tc-mxmT sample parity vs oracle ({int(Mp_s[-1])} mask entries): {'OK' if ok else 'MISMATCH'}")
            if not ok:
                raise SystemExit("bench: mxmT result differs from the oracle")
            line["verified"] = True
        if not args.no_cpu_baseline and pyoracle.have_reference():
            ref = pyoracle.Reference()
            sdeg = (Mp_s[1:] - Mp_s[:-1]).to(torch.int64)
            srow_of = torch.repeat_interleave(torch.arange(n, device="cuda"), sdeg)
            swork = int((deg[srow_of] + deg[Mj_s.to(torch.int64)]).sum())
            t0 = time.perf_counter()
            ref.mxmT_masked("INT", n, Mh, Lh, Lh, "MULT", "PLUS", "GTZERO", 0, n_cols=n, inner=n)
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(swork / dt / 1e9, 4), "unit": "Gsteps/s", "cores": 1, "kind": "reference",
                                    "sample": f"mask restricted to every 64th row ({int(Mp_s[-1])} entries, {swork} merge steps), one full call incl. "
                                              "COO->LIL conversion of the three operands", "host_cpus": os.cpu_count(), "ms": round(dt * 1e3, 1)}
    return line


# ---------------------------------------------------------------------------------------------------------------- mxm
def mxm(args):
    from oracle import pyoracle
    from spla_amd import graphgen, kernels
    scale = getattr(args, "mxm_scale", 0) or (18 if args.scale == 22 else args.scale)
    n, Ap, Aj = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda")
    nnz = int(Ap[-1])
    g = torch.Generator(device="cuda")
    g.manual_seed(args.seed + 9)
    Ax = (torch.rand(nnz, generator=g, device="cuda") * 0.5 + 0.25).contiguous()
    A = (Ap, Aj, Ax)
    out = {}

    # the timed region is the C call (blocking: it returns when the result is complete); copying the 10 GB result into torch
    # tensors and releasing the library's arrays is the Python wrapper's business and stays outside
    steps, warm = max(1, min(args.steps, 5)), 1
    walls = []
    for it in range(warm + steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        raw, P = kernels.mxm("FLOAT", n, n, A, A, "MULT", "PLUS", 0.0, adopt=False)
        dt = time.perf_counter() - t0
        if it >= warm:
            walls.append(dt)
        if it + 1 < warm + steps:
            raw.free()
    wall, ms = float(np.sum(walls)), float(np.mean(walls)) * 1e3
    out["P"] = P
    Rp, Rj, Rx = raw.adopt("FLOAT")
    nnz_c = int(Rj.numel())
    _log(f"[bench] mxm: R-MAT scale {scale}: n={n} nnz(A)={nnz} products={P} nnz(C)={nnz_c}")
    alg = 8 * (2 * nnz + nnz_c) + 4 * 3 * (n + 1)
    args_steps = args.steps
    args.steps = steps
    line = _line("SpGEMM (mxm) G products/s", "Gproducts/s", P / (wall / steps) / 1e9, args, wall / steps * 1e3, "f32",
                 {"workload": f"exec_mxm A x A, MULT/PLUS fp32, init 0; R-MAT scale {scale} ef {args.edge_factor} (BASELINE configs[4] names scale 20: its result holds more than "
                              "2^32 entries, which the reference's 32-bit row pointers (include/spla/config.hpp:56, `uint`) cannot index — the call refuses it, "
                              "tests/test_mxm_gpu.py::test_mxm_refuses_a_result_beyond_32bit_row_pointers; scale 19, 3.53 G result entries, is the largest that fits)",
                  "n": n, "nnz_A": nnz, "products": P, "nnz_C": nnz_c,
                  "form": "row classes (mxm_spa.hip): rows of > 4096 products keep the result row's accumulators in LDS (Gustavson; symbolic pass, then the ranked form "
                          "for rows whose entries fit one block and column blocks for the hubs), lighter rows through ESC; a dense leading corner would run on MFMA "
                          "(none in R-MAT: the 1 K x 1 K hub corner is ~23 % full); timed: the blocking C call", "row_classes": _mxm_info()},
                 _roof(alg, ms, "mxm_spa (symbolic + numeric) + ESC for the light rows", pmc="mxm" if scale == 18 else "mxm_scale19" if scale == 19 else None))
    args.steps = args_steps
    if not args.no_verify or not args.no_cpu_baseline:
        del Rp, Rj, Rx  # (scale 19: 28 GB)
        torch.cuda.empty_cache()
        srows = torch.arange(0, n, 256, device="cuda")
        Sp, Sj, Sx = _row_sample_csr(Ap, Aj, Ax, srows)
        (gp, gj, gx), sprod = kernels.mxm("FLOAT", n, n, (Sp, Sj, Sx), A, "MULT", "PLUS", 0.0)
        Ah = tuple(t.cpu().numpy() for t in A)
        Sh = (Sp.cpu().numpy(), Sj.cpu().numpy(), Sx.cpu().numpy())
        if not args.no_verify:
            want = pyoracle.Oracle().mxm("FLOAT", n, n, Sh, Ah, "MULT", "PLUS", 0.0)
            ok = np.array_equal(want[0], gp.cpu().numpy().view(np.uint32)) and np.array_equal(want[1], gj.cpu().numpy().view(np.uint32)) and \
                np.allclose(want[2], gx.cpu().numpy(), rtol=1e-5, atol=0)
            _log(f"[bench] mxm sample parity vs oracle (every 256th row of A, {sprod} products): {'OK' if ok else 'MISMATCH'}")
            if not ok:
                raise SystemExit("bench: mxm result differs from the oracle")
            line["verified"] = True
        if not args.no_cpu_baseline and pyoracle.have_reference():
            # the reference's mxm is O(rows x n_cols) per call (cpu_mxm.hpp:93,109): time it on a 2^12-row, 2^12-column corner
            m = min(n, 4096)
            keep = Aj[: int(Ap[m])] < m
            rows = torch.repeat_interleave(torch.arange(m, device="cuda"), (Ap[1:m + 1] - Ap[:m]).to(torch.int64))
            key = rows[keep] * m + Aj[: int(Ap[m])][keep].to(torch.int64)
            Cp, Cj = graphgen.keys_to_csr(key, m, m)
            Cx = Ax[: int(Ap[m])][keep].contiguous()
            Ch = (Cp.cpu().numpy(), Cj.cpu().numpy(), Cx.cpu().numpy())
            cprod = int(sum(int(Ch[0][c + 1]) - int(Ch[0][c]) for c in Ch[1]))
            ref = pyoracle.Reference()
            t0 = time.perf_counter()
            ref.mxm("FLOAT", m, m, Ch, Ch, "MULT", "PLUS", 0.0)
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(cprod / dt / 1e9, 5), "unit": "Gproducts/s", "cores": 1, "kind": "reference",
                                    "sample": f"leading {m} x {m} corner of A ({int(Cp[-1])} nnz, {cprod} products): the reference refills and rescans an "
                                              "n_cols-wide accumulator per row, so it is timed at reduced size; one full call incl. conversions",
                                    "host_cpus": os.cpu_count(), "ms": round(dt * 1e3, 1)}
    return line


# ------------------------------------------------------------------------------------------------------- algorithms
def algorithms(args):
    """End-to-end graph algorithms over exec_* on the HIP accelerator (spla_amd/csrc/host/algorithms.cpp)."""
    from oracle import pyoracle
    from spla_amd import api
    scale = args.scale if args.scale != 22 else 20
    n, Ap, Aj = _sym_rmat(scale, args.edge_factor, args.seed)
    nnz = int(Ap[-1])
    Ap_h, Aj_h = Ap.cpu().numpy().astype(np.uint32), Aj.cpu().numpy().astype(np.uint32)
    I_h = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap_h.astype(np.int64)))
    deg = np.diff(Ap_h.astype(np.int64))
    src = int(np.argmax(deg))
    rng = np.random.default_rng(args.seed)
    lo, hi = np.minimum(I_h, Aj_h).astype(np.int64), np.maximum(I_h, Aj_h).astype(np.int64)
    W_h = (((lo * 2654435761 + hi * 40503) % 255) + 1).astype(np.float32)  # symmetric weights 1..255
    P_h = (np.float32(0.85) / deg[Aj_h].astype(np.float32)).astype(np.float32)
    keep = Aj_h < I_h
    Lp_h = np.zeros(n + 1, np.uint32)
    np.cumsum(np.bincount(I_h[keep], minlength=n), out=Lp_h[1:])
    _log(f"[algorithms] symmetrised R-MAT scale {scale}: n={n} nnz={nnz} max degree {int(deg.max())} source {src}")
    oracle = pyoracle.Oracle()
    res = {}

    def timed(name, fn, reps):
        for _ in range(3):  # first calls: host COO -> CSR conversion, upload, plans; the matrix's layout copies arrive with its second / third product
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        _log(f"[algorithms] {name}: {1e3 * float(np.median(ts)):.2f} ms (median of {reps})")
        return out, float(np.median(ts))

    A = api.Matrix.from_lists(I_h, Aj_h, np.ones(nnz, np.int32), (n, n), api.INT)
    want = oracle.bfs(n, Ap_h, Aj_h, src)
    reached_edges = int(deg[want > 0].sum())
    for label, desc in (("push-pull (default)", None), ("pull only", api.Descriptor(traversal_mode=api.Descriptor.PULL)), ("push only", api.Descriptor(traversal_mode=api.Descriptor.PUSH))):
        lv, dt = timed(f"bfs {label}", lambda d=desc: api.bfs(A, src, d), 21)
        ok = bool(np.array_equal(lv.to_dense() if n <= (1 << 16) else _dense_i(lv, n), want))
        res[f"bfs {label}"] = {"ms": round(dt * 1e3, 3), "gteps": round(reached_edges / dt / 1e9, 3), "levels": int(want.max()), "verified": ok}
    del A
    Wm = api.Matrix.from_lists(I_h, Aj_h, W_h, (n, n), api.FLOAT)
    want = oracle.sssp(n, Ap_h, Aj_h, W_h, src)
    dv, dt = timed("sssp push-pull", lambda: api.sssp(Wm, src), 11)
    res["sssp push-pull (default)"] = {"ms": round(dt * 1e3, 3), "verified": bool(np.array_equal(_dense_f(dv, n).view(np.uint32), want.view(np.uint32)))}
    del Wm
    Pm = api.Matrix.from_lists(I_h, Aj_h, P_h, (n, n), api.FLOAT)
    pv, dt = timed("pr alpha 0.85 eps 1e-5", lambda: api.pr(Pm, 0.85, 1e-5), 11)
    want = oracle.pr(n, Ap_h, Aj_h, P_h, 0.85, 1e-5)
    got = _dense_f(pv, n, fill=0.0)
    res["pr alpha 0.85 eps 1e-5"] = {"ms": round(dt * 1e3, 3), "max_abs_vs_oracle": float(np.abs(got - want).max()), "verified": bool(np.allclose(got, want, rtol=1e-3, atol=1e-8))}
    del Pm
    Lm = api.Matrix.from_lists(I_h[keep], Aj_h[keep], np.ones(int(keep.sum()), np.int32), (n, n), api.INT)
    (cnt, B), dt = timed("tc", lambda: api.tc(Lm), 3)
    res["tc"] = {"ms": round(dt * 1e3, 3), "triangles": int(cnt)}
    if scale <= 16:
        res["tc"]["verified"] = bool(cnt == oracle.tc(n, Lp_h, Aj_h[keep])[0])
    del Lm, B

    # reference CPU backend, same graph, bfs only (1 core; bounded: one call)
    cpu = None
    try:
        ref = pyoracle.Reference()
        t0 = time.perf_counter()
        lv = ref.bfs(n, Ap_h, Aj_h, src)
        dt = time.perf_counter() - t0
        cpu = {"value": round(reached_edges / dt / 1e9, 4), "unit": "GTEPS", "cores": 1, "kind": "reference",
               "sample": "spla_Algorithm_bfs (push-pull) of the genuine reference on the same graph, one call incl. its format conversions",
               "host_cpus": os.cpu_count(), "ms": round(dt * 1e3, 1), "matches_hip": bool(np.array_equal(lv, oracle.bfs(n, Ap_h, Aj_h, src)))}
    except Exception as e:  # oracle/_ref absent
        _log("[algorithms] reference baseline skipped:", e)
    main = res["bfs push-pull (default)"]
    line = _line("BFS end-to-end GTEPS (spla_Algorithm_bfs on the HIP accelerator)", "GTEPS", main["gteps"], args, main["ms"], "int32",
                 {"workload": f"bfs / sssp / pr / tc through the operator API; symmetrised R-MAT scale {scale} ef {args.edge_factor}", "n": n, "nnz": nnz, "source": src},
                 {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                  "kernel": "whole algorithm (host-synchronised iterations): no single-kernel byte model applies"},
                 {"algorithms": res, "verified": all(v.get("verified", True) for v in res.values())})
    if cpu:
        line["cpu_baseline"] = cpu
    return line


# ---------------------------------------------------------------------------------------------------------- transpose
def transpose(args):
    """R = AINV(M)^T through the kernel-level C ABI; algorithmic bytes = read M + write R = 2 * (8 nnz + 4 (n + 1))."""
    from oracle import pyoracle
    from spla_amd import _lib, graphgen
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import type_id, unop_id
    scale = int(os.environ.get("SPLA_TRANSPOSE_SCALE", "22"))
    n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
    nnz = int(Aj.numel())
    Ax = (torch.arange(nnz, device="cuda", dtype=torch.int32) % 1000003)
    Rp = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    Rj, Rx = torch.empty(nnz, dtype=torch.int32, device="cuda"), torch.empty(nnz, dtype=torch.int32, device="cuda")
    tid, op = type_id("INT"), unop_id("AINV")

    def step():
        _lib.call("splahip_m_transpose", tid, n, n, nnz, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(Rp), _ptr(Rj), _ptr(Rx), op, current_stream_ptr())

    wall, ms = _time_calls(step, args.steps, args.warmup)
    med = float(np.median(ms))
    # parity on a bounded sample: the first rows of M against the oracle, and (M^T)^T == M in full
    Bp, Bj, Bx = torch.empty_like(Rp), torch.empty_like(Rj), torch.empty_like(Rx)
    _lib.call("splahip_m_transpose", tid, n, n, nnz, _ptr(Rp), _ptr(Rj), _ptr(Rx), _ptr(Bp), _ptr(Bj), _ptr(Bx), op, current_stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(Bp, Ap) and torch.equal(Bj, Aj) and torch.equal(Bx, Ax), "transpose twice is not the identity"
    rows = min(n, 1 << 16)
    hp = Ap[:rows + 1].cpu().numpy().view(np.uint32)
    k = int(hp[-1])
    hj, hx = Aj[:k].cpu().numpy().view(np.uint32), Ax[:k].cpu().numpy()
    orc = pyoracle.Oracle()
    t0 = time.perf_counter()
    want = orc.m_transpose("INT", rows, n, hp, hj, hx, "AINV")
    cpu_s = time.perf_counter() - t0
    sp, sj, sx = torch.empty(n + 1, dtype=torch.int32, device="cuda"), torch.empty(max(k, 1), dtype=torch.int32, device="cuda"), torch.empty(max(k, 1), dtype=torch.int32, device="cuda")
    dp = torch.from_numpy(hp.view(np.int32)).cuda()
    _lib.call("splahip_m_transpose", tid, rows, n, k, _ptr(dp), _ptr(Aj), _ptr(Ax), _ptr(sp), _ptr(sj), _ptr(sx), op, current_stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(sp.cpu().numpy().view(np.uint32), want[0]) and np.array_equal(sj[:k].cpu().numpy().view(np.uint32), want[1]) \
        and np.array_equal(sx[:k].cpu().numpy(), want[2]), "transpose differs from the oracle"
    alg = 2 * (8 * nnz + 4 * (n + 1))
    roof = _roof(alg, med, "rocprim radix sort passes + transpose_payload / emit / offsets")
    _log(f"[transpose] n={n} nnz={nnz}: {med:.3f} ms per call ({nnz / med / 1e6:.2f} G entries/s), oracle {k / cpu_s / 1e6:.1f} M entries/s on one core")
    return _line("transpose_gentries_per_s", "Gentries/s", nnz / (med * 1e-3) / 1e9, args, med, "i32",
                 {"workload": f"exec_m_transpose AINV int32, R-MAT scale {scale} edge factor 16 ({nnz} entries)"}, roof,
                 {"cpu_baseline": {"value": round(k / cpu_s / 1e9, 4), "unit": "Gentries/s", "cores": 1, "kind": "port",
                                   "sample": f"first {rows} rows ({k} entries), oracle_m_transpose"}})


def _dense_i(vec, n):
    """dense read-back of an INT vector through the C ABI's bulk read (absent entries = 0)"""
    I, X = vec.read()
    out = np.zeros(n, np.int32)
    out[I] = X
    return out


def _dense_f(vec, n, fill=None):
    I, X = vec.read()
    out = np.full(n, np.float32(np.finfo(np.float32).max) if fill is None else np.float32(fill), np.float32)
    out[I] = X
    return out


# ---------------------------------------------------------------------------------------------------------------- vxm push, mid-size frontier
def vxm_push(args):
    """exec_vxm_masked in its push form on a mid-size frontier: 10^6 stored entries of v over R-MAT scale 22, fp32 PLUS / MIN products folded per
    output column in the reference's order (expand, stable radix sort by output index, left fold of every run: bit-exact), and the int32 BAND / BOR step of a traversal
    (order-free: memory-side atomics + compaction).  Throughput in products; algorithmic bytes = 12 B per product (Aj, Ax, the product's key / value once) + the vector
    and the mask."""
    from oracle import pyoracle
    from spla_amd import graphgen, kernels
    scale = 22 if args.scale == 22 else args.scale
    n, Ap, Aj = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda")
    nnz = int(Ap[-1])
    g = torch.Generator(device="cuda")
    g.manual_seed(args.seed + 31)
    k = min(1_000_000, n // 4)
    vi = torch.sort(torch.randperm(n, generator=g, device="cuda")[:k]).values.to(torch.int32)
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    products = int(deg[vi.to(torch.int64)].sum())
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    lines = []
    for t, mul, add, sel, Ax, vx in (("FLOAT", "PLUS", "MIN", "ALWAYS", torch.randint(1, 256, (nnz,), generator=g, device="cuda").to(torch.float32),
                                      torch.randint(0, 5000, (k,), generator=g, device="cuda").to(torch.float32)),
                                     ("INT", "BAND", "BOR", "EQZERO", torch.ones(nnz, dtype=torch.int32, device="cuda"), torch.ones(k, dtype=torch.int32, device="cuda"))):
        m = mask if t == "FLOAT" else mask.to(torch.int32)
        out = {}

        def step():
            out["r"] = kernels.vxm_push(t, n, n, Ap, Aj, Ax, m, vi, vx, mul, add, sel)

        wall, per = _time_calls(step, args.steps, args.warmup)
        ms = float(np.mean(per))
        alg = 12 * products + 8 * k + 4 * n
        e = {"name": f"{t} {mul}/{add}, {k} stored entries of v ({products} products)", "ms_per_call": round(ms, 4), "gproducts_per_s": round(products / (ms * 1e-3) / 1e9, 2),
             "nnz_result": int(out["r"][0].numel()), "roofline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if not args.no_verify:  # the first 20000 stored entries of v against the oracle (a prefix of the frontier is a valid frontier)
            kk = 20000
            ri, rx = kernels.vxm_push(t, n, n, Ap, Aj, Ax, m, vi[:kk], vx[:kk], mul, add, sel)
            wi, wx = pyoracle.Oracle().vxm_masked(t, n, n, Ap.cpu().numpy(), Aj.cpu().numpy(), Ax.cpu().numpy(), m.cpu().numpy(), vi[:kk].cpu().numpy(), vx[:kk].cpu().numpy(), mul, add, sel)
            ok = bool(np.array_equal(ri.cpu().numpy().view(np.uint32), wi) and np.array_equal(rx.cpu().numpy().view(np.uint32), np.asarray(wx).view(np.uint32)))
            if not ok:
                raise SystemExit("bench: vxm push differs from the oracle")
            e["verified"] = True
        lines.append(e)
    f = lines[0]
    alg = 12 * products + 8 * k + 4 * n
    line = _line("vxm push G products/s (mid-size frontier)", "Gproducts/s", f["gproducts_per_s"], args, f["ms_per_call"], "f32",
                 {"workload": f"exec_vxm_masked push form, fp32 PLUS/MIN, {k} stored entries of v over R-MAT scale {scale} ef {args.edge_factor}", "n": n, "nnz": nnz, "products": products},
                 _roof(alg, f["ms_per_call"], "push: expand + radix sort by output index (rocPRIM) + in-order fold of every run"))
    line["forms"] = lines
    line["verified"] = all(x.get("verified", False) for x in lines) if not args.no_verify else None
    return line


def mxm_dense(args):
    """The MFMA sub-path of exec_mxm measured (north star: "MFMA used only ... where a row-bucket degenerates to a dense tile"): R-MAT scale 16 with a
    planted fully dense H x H block on the diagonal, away from the origin (rows and columns [2H, 3H), H = 4096 unless --scale says otherwise), A x A fp32 MULT / PLUS.
    Three runs of the same call: the window capped at 1024 (the library's cap until round 6), the window allowed the whole block (SPLAHIP_MXM_DENSE_H = H = the default now), and no window
    (SPLAHIP_MXM_DENSE=0: every product through the sparse row classes).  Reported: the window found, the time of its MFMA GEMM (values; a second GEMM of the same shape
    counts the products per cell) against the dense fp32 MFMA peak, end-to-end products/s of each run, parity against the oracle on a row sample (rows through the block
    included) and agreement of the runs' structure."""
    import ctypes as C
    from oracle import pyoracle
    from spla_amd import _lib, graphgen, kernels
    scale, H = 16, 4096
    n, Ap0, Aj0 = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda")
    lo, hi_ = 2 * H, 3 * H
    rows0 = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap0[1:] - Ap0[:-1]).to(torch.int64))
    blk = torch.arange(lo, hi_, device="cuda", dtype=torch.int64)
    keys = torch.unique(torch.cat([rows0 * n + Aj0.to(torch.int64), (blk[:, None] * n + blk[None, :]).reshape(-1)]), sorted=True)
    del rows0
    Ap, Aj = graphgen.keys_to_csr(keys, n, n)
    del keys
    nnz = int(Ap[-1])
    g = torch.Generator(device="cuda")
    g.manual_seed(args.seed + 19)
    Ax = (torch.rand(nnz, generator=g, device="cuda") * 0.5 + 0.25).contiguous()
    A = (Ap, Aj, Ax)

    def one(env):
        for k, v in env.items():
            os.environ[k] = v
        _lib.call("splahip_env_refresh")
        try:
            walls, P, win, gemm_ms, raw = [], 0, None, 0.0, None
            for it in range(3):
                if raw is not None:
                    raw.free()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                raw, P = kernels.mxm("FLOAT", n, n, A, A, "MULT", "PLUS", 0.0, adopt=False)
                dt = time.perf_counter() - t0
                if it:
                    walls.append(dt)
                w = (C.c_uint32 * 6)()
                _lib.call("splahip_mxm_last_dense", w)
                win = [int(x) for x in w]
                ms = C.c_float(0.0)
                _lib.call("splahip_mxm_last_dense_ms", C.byref(ms))
                gemm_ms = float(ms.value)
            return float(np.mean(walls)) * 1e3, P, win, gemm_ms, raw
        finally:
            for k in env:
                os.environ.pop(k, None)
            _lib.call("splahip_env_refresh")

    res = {}
    runs = {"window <= 1024 (the library's cap until round 6)": {"SPLAHIP_MXM_DENSE_TIME": "1", "SPLAHIP_MXM_DENSE_H": "1024"},
            f"window <= {H} (the whole planted block; the library default)": {"SPLAHIP_MXM_DENSE_TIME": "1", "SPLAHIP_MXM_DENSE_H": str(H)},
            "no window (all sparse)": {"SPLAHIP_MXM_DENSE": "0"}}
    structure = None
    for name, env in runs.items():
        ms, P, win, gemm_ms, raw = one(env)
        e = {"ms_per_call": round(ms, 2), "products": P, "gproducts_per_s": round(P / (ms * 1e-3) / 1e9, 2), "window_i0_k0_j0_hi_hk_hj": win}
        if win[3]:
            flops = 2.0 * win[3] * win[4] * win[5]
            e["window_products"] = int(win[3]) * int(win[4]) * int(win[5])
            if gemm_ms > 0:
                # the event pair brackets TWO GEMMs of the same shape (values; product counts): the rate is quoted per GEMM
                e["mfma_gemm_ms_each"] = round(gemm_ms / 2, 4)
                e["mfma_gemm_tflops"] = round(flops / (gemm_ms / 2 * 1e-3) / 1e12, 2)
                e["mfma_frac_of_fp32_peak"] = round(flops / (gemm_ms / 2 * 1e-3) / 1e12 / 157.3, 4)
        Rp = raw.Rp.clone()
        if structure is None:
            structure = Rp
            e["same_structure_as_first_run"] = True
        else:
            e["same_structure_as_first_run"] = bool(torch.equal(structure, Rp))
        res[name] = e
        _log(f"[bench] mxm-dense {name}: {e}")
        if name.startswith(f"window <= {H} "):
            best = raw  # (checked against the oracle below)
        else:
            raw.free()
    # parity of the full-window run on a row sample: every 64th row plus 64 rows through the planted block
    line_ok = None
    if not args.no_verify:
        Rp64 = best.Rp.to(torch.int64) & 0xFFFFFFFF
        Rj, Rx = best.adopt("FLOAT")[1:]
        srows = torch.unique(torch.cat([torch.arange(0, n, 64, device="cuda"), torch.arange(lo, hi_, 64, device="cuda") + 7]))
        Sp, Sj, Sx = _row_sample_csr(Ap, Aj, Ax, srows)
        Ah = tuple(t.cpu().numpy() for t in A)
        wp, wj, wx = pyoracle.Oracle().mxm("FLOAT", n, n, (Sp.cpu().numpy(), Sj.cpu().numpy(), Sx.cpu().numpy()), Ah, "MULT", "PLUS", 0.0)  # (same shape, unsampled rows empty)
        lens = (Rp64[1:] - Rp64[:-1])[srows]
        Op = torch.zeros(srows.numel() + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(lens, 0, out=Op[1:])
        orep = torch.repeat_interleave(torch.arange(srows.numel(), device="cuda"), lens)
        pos = Rp64[srows][orep] + (torch.arange(int(Op[-1]), device="cuda") - Op[:-1][orep])
        same_structure = bool(np.array_equal(lens.cpu().numpy(), np.diff(wp.astype(np.int64))[srows.cpu().numpy()]) and np.array_equal(Rj[pos].cpu().numpy().view(np.uint32), wj))
        # values: an entry of the planted block is a sum of >= 4096 products — the reference's sequential fp32 fold is itself up to ~1e-4 away from the sum there, so,
        # as for the long rows of the SpMV (tests/test_configs_gpu.py: _assert_fp32_sum_parity), every entry is held to 1e-5 relative PLUS the oracle's own distance
        # from an fp64 evaluation of the same products, and the share inside plain 1e-5 is reported
        import scipy.sparse as sp
        S64 = sp.csr_matrix((Sx.cpu().numpy().astype(np.float64), Sj.cpu().numpy().astype(np.int64), Sp.cpu().numpy().astype(np.int64)), shape=(n, n))
        A64 = sp.csr_matrix((Ah[2].astype(np.float64), Ah[1].astype(np.int64), Ah[0].astype(np.int64)), shape=(n, n))
        E = (S64 @ A64).tocsr()
        E.sort_indices()
        got64, want64 = Rx[pos].cpu().numpy().astype(np.float64), wx.astype(np.float64)
        exact_ok = same_structure and E.nnz == len(wj) and np.array_equal(E.indices.astype(np.uint32), wj)
        ex = E.data if exact_ok else want64
        diff = np.abs(got64 - want64) if same_structure else np.array([np.inf])
        plain = diff <= 1e-5 * np.abs(want64) if same_structure else np.array([False])
        line_ok = bool(same_structure and exact_ok and np.all(diff <= 1e-5 * np.abs(want64) + 1.01 * np.abs(want64 - ex) + 1e-30))
        parity = {"rows": int(srows.numel()), "entries": int(len(wj)), "same_structure": same_structure, "within_plain_1e-5": float(plain.mean()),
                  "max_rel_vs_oracle": float(np.max(diff / np.maximum(np.abs(want64), 1e-30))) if same_structure else None,
                  "max_rel_vs_fp64": {"hip": float(np.max(np.abs(got64 - ex) / np.maximum(np.abs(ex), 1e-30))), "reference_sequential_fold": float(np.max(np.abs(want64 - ex) / np.maximum(np.abs(ex), 1e-30)))} if exact_ok else None}
        _log(f"[bench] mxm-dense parity vs oracle: {parity}: {'OK' if line_ok else 'MISMATCH'}")
        if not line_ok:
            raise SystemExit("bench: mxm-dense result differs from the oracle")
    else:
        best.free()
    full = res[f"window <= {H} (the whole planted block; the library default)"]
    line = _line("SpGEMM (mxm) with a dense block on the matrix cores, G products/s", "Gproducts/s", full["gproducts_per_s"], args, full["ms_per_call"], "f32",
                 {"workload": f"exec_mxm A x A, MULT/PLUS fp32: R-MAT scale {scale} ef {args.edge_factor} + a planted fully dense {H} x {H} block at rows / columns [{lo}, {hi_})",
                  "n": n, "nnz_A": nnz, "products": full["products"], "runs": res,
                  "mfma": "v_mfma_f32_32x32x2_f32, LDS-staged double-buffered K panels of 32, a 64 x 64 tile of C per four-wave workgroup (mxm_spa.hip: dense_gemm_mfma_kernel); "
                          "dense fp32 MFMA peak 157.3 TFLOP/s"},
                 {"bound": "mfma", "achieved": full.get("mfma_gemm_tflops"), "peak": 157.3, "unit": "TFLOP/s", "frac": full.get("mfma_frac_of_fp32_peak"), "traffic": None,
                  "kernel": "dense_gemm_mfma_kernel (the window's values GEMM)"})
    if line_ok is not None:
        line["verified"], line["parity"] = line_ok, parity
    return line


def mxm_scale19(args):
    """exec_mxm A x A at R-MAT scale 19: 3.53 G result entries, the largest result of this family the reference's 32-bit row pointers can index."""
    import copy
    b = copy.copy(args)
    b.mxm_scale = 19
    b.steps = min(b.steps, 3)
    return mxm(b)


WORKLOADS = {"vxm-push": vxm_push, "pagerank-vxm": pagerank_vxm, "sssp-step": sssp_step, "tc-mxmT": tc_mxmT, "mxm": mxm, "mxm-19": mxm_scale19, "mxm-dense": mxm_dense, "algorithms": algorithms,
             "transpose": transpose}
