"""bench.py's default workload since round 3: the NORTH-STAR configuration — exec_mxv_masked, fp32 MULT / PLUS over ALL rows (op_select ALWAYS)
of a ~100 M-nnz scale-free CSR: R-MAT scale 23 (edge factor 16, self loops + duplicates removed: 131 M nnz, n = 8.4 M), raw labels.

    N = 1   the matrix with its column-ordered-blocks copy (splahip_csr_plan_attach_cob) resident in HBM, vectors resident;
            a step = one splahip_mxv_masked call.  `value` = nnz / step time (GTEPS), `roofline` = SURVEY 8d's algorithmic bytes
            (8 nnz + 4 (n + 1) + 4 n mask + 4 n r + 4 n v) / the call's HIP-event duration / 8 TB/s, `cpu_baseline` = the genuine reference
            exec_mxv_masked (oracle/_ref) on every s-th row of the SAME matrix on one host core, its rows also checked against the HIP result.
    N > 1   weak scaling: R-MAT scale 23 + log2(N), owner-sorted labels, 1-D nnz-balanced row blocks, every step through the C-ABI dist layer
            (splahip_dist_mxv_masked: RCCL all-gather of the dense fp32 vector on the communicator's stream, overlapped with the local-column part;
            SPLA_ALLGATHER=direct selects grouped send/recv per peer).  value = global nnz / max-over-ranks step time.
"""
from __future__ import annotations

import json
import math
import os
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

_T0 = time.time()

HBM_PEAK_GBS = 8000.0
ROOT = Path(__file__).resolve().parent


def _log(*a):
    import sys
    print(*a, file=sys.stderr, flush=True)


def alg_bytes(nnz, n_rows, n_cols):
    return 8 * nnz + 4 * (n_rows + 1) + 4 * n_rows + 4 * n_rows + 4 * n_cols


def time_calls(fn, steps, warmup, world=1):
    """Contract timing: W untimed steps, exactly K steps between barrier + synchronize, wall = max over ranks; per-call HIP events on the launch stream."""
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


def copy_ceiling_gbs(n_words=1 << 28, reps=10):
    """GB/s (read + write) of a 1 GiB device-to-device copy: the streaming ceiling SURVEY 8d asks to report beside the nominal peak."""
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


def read_stream_gbs(n_bytes=1 << 30, reps=10):
    """GB/s of this library's own read-only stream over 1 GiB (splahip_read_stream_gbs: 16 bytes per lane, non-temporal, a 1024-thread workgroup per CU): what
    a kernel that only had to READ its bytes reaches on this GPU in this run — the honest measured denominator for a read-dominated kernel (VERDICT r5 item 7;
    a device-to-device copy, 4.6 - 4.8 TB/s of read + write, is the lower of the two and flatters)."""
    import ctypes as C
    from spla_amd import _lib
    g = C.c_double(0.0)
    _lib.call("splahip_read_stream_gbs", C.c_size_t(n_bytes), reps, C.byref(g), None)
    return float(g.value)


def sample_rows(Ap, Aj, Ax, stride):
    """every stride-th row of a device CSR as host arrays (Ap_s, Aj_s, Ax_s) + the sampled row ids (device)"""
    n = Ap.numel() - 1
    rows = torch.arange(0, n, stride, device=Ap.device)
    Ap64 = Ap.to(torch.int64)
    deg = (Ap64[1:] - Ap64[:-1])[rows]
    Aps = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=Ap.device)
    torch.cumsum(deg, 0, out=Aps[1:])
    rep = torch.repeat_interleave(torch.arange(rows.numel(), device=Ap.device), deg)
    src = Ap64[rows][rep] + (torch.arange(int(Aps[-1]), device=Ap.device) - Aps[:-1][rep])
    return rows, Aps.to(torch.int32).cpu().numpy().view(np.uint32), Aj[src].cpu().numpy().view(np.uint32), Ax[src].cpu().numpy()


def rel_err(got, want):
    want64 = want.astype(np.float64)
    return float(np.max(np.abs(got.astype(np.float64) - want64) / np.maximum(np.abs(want64), 1e-30))) if got.size else 0.0


def cpu_reference_sample(t, Ap, Aj, Ax, mask, v, r, mul, add, sel, init, sample_nnz, rtol):
    """The reference CPU path (genuine reference build when present, else the plain-C port) on every s-th row of the same workload, 1 core;
    the sampled rows are compared with the HIP result (bit-exact, or rtol for an fp32 sum)."""
    from oracle import pyoracle
    nnz = int(Ap[-1])
    n_cols = v.numel()
    stride = max(1, int(round(nnz / max(1, sample_nnz))))
    rows, Ap_s, Aj_s, Ax_s = sample_rows(Ap, Aj, Ax, stride)
    m, nnz_s = int(rows.numel()), int(Ap_s[-1])
    tid = {"INT": 0, "UINT": 1, "FLOAT": 2}[t]
    np_t = {"INT": np.int32, "UINT": np.uint32, "FLOAT": np.float32}[t]
    mask_s, v_h = mask[rows].cpu().numpy(), v.cpu().numpy()
    times, first = [], None
    if pyoracle.have_reference():
        ref = pyoracle.Reference()
        kind = "reference"
        M = ref._matrix(tid, m, n_cols, Ap_s, Aj_s, Ax_s)
        hm, hv, hr = ref._dense_vector(tid, mask_s), ref._dense_vector(tid, v_h), ref._empty_vector(tid, m)
        hi = ref._scalar(init, tid)
        ops = (ref._op("Binary", mul, tid), ref._op("Binary", add, tid), ref._op("Select", sel, tid))
        for it in range(4):
            t0 = time.perf_counter()
            assert ref.lib.spla_Exec_mxv_masked(hr, hm, M, hv, ops[0], ops[1], ops[2], hi, None, None) == 0
            dt = time.perf_counter() - t0
            if it:
                times.append(dt)
            else:
                first = dt
        got = np.empty(m, np.uint32)
        ref.lib.refshim_vector_dense_read(hr, got.ctypes.data_as(pyoracle._u32p))
        got = got.view(np_t)
        ref._unref(M, hm, hv, hr, hi)
    else:
        orc = pyoracle.Oracle()
        kind = "port"
        for it in range(4):
            t0 = time.perf_counter()
            got = orc.mxv_masked(t, m, n_cols, Ap_s, Aj_s, Ax_s, mask_s, v_h, mul, add, sel, init)
            if it:
                times.append(time.perf_counter() - t0)
    hip = r[rows].cpu().numpy()
    if rtol:  # fp32 sum: the bar plus the reference's own distance from an fp64 fold of the sampled rows (see run(): hub rows)
        err = rel_err(hip, got)
        ex = np.bincount(np.repeat(np.arange(m), np.diff(Ap_s.astype(np.int64))), weights=Ax_s.astype(np.float64) * v_h[Aj_s.astype(np.int64)].astype(np.float64), minlength=m)
        g64 = got.astype(np.float64)
        same = bool(np.all(np.abs(hip.astype(np.float64) - g64) <= rtol * np.abs(g64) + 1.01 * np.abs(g64 - ex) + 1e-30))
    else:
        err = None
        same = bool(np.array_equal(got.view(np.uint32), hip.view(np.uint32)))
    med = float(np.median(times))
    _log(f"[bench] cpu baseline ({kind}): every {stride}th row: {m} rows / {nnz_s} nnz, median {med * 1e3:.1f} ms, HIP matches: {same}" + (f" (max rel {err:.2e})" if err is not None else ""))
    if not same:
        raise SystemExit("bench: HIP result differs from the CPU reference on the sample")
    out = {"value": round(nnz_s / med / 1e9, 4), "unit": "GTEPS", "cores": 1, "kind": kind,
           "sample": f"every {stride}th row of the same matrix: {m} rows, {nnz_s} nnz, same v; median of 3 steady-state exec_mxv_masked calls after 1 warm-up"
                     + (f" (first call incl. COO->LIL conversion {first * 1e3:.0f} ms)" if first else ""),
           "host_cpus": os.cpu_count(), "ms": round(med * 1e3, 2), "matches_hip": same}
    if err is not None:
        out["max_rel_vs_hip"] = err
    return out


def pmc_traffic(key):
    """per-call fabric traffic of this workload from the committed counter pass (separate rocprofv3 --pmc runs; profiles/r03_pmc_<key>.json)"""
    f = ROOT / "profiles" / f"r06_pmc_{key}.json"
    if not f.exists():
        f = ROOT / "profiles" / f"r05_pmc_{key}.json"
    if not f.exists():
        f = ROOT / "profiles" / f"r04_pmc_{key}.json"
    if not f.exists():  # (the kernels of this call did not change since round 3's pass)
        f = ROOT / "profiles" / f"r03_pmc_{key}.json"
    try:
        return json.loads(f.read_text()).get("_per_call"), f.name
    except (OSError, ValueError):
        return None, f.name


def run(args, rank, world):
    from oracle import pyoracle
    from spla_amd import graphgen, kernels, sharding
    scale = (args.scale if args.scale else 23) + int(round(math.log2(world)))
    pm = (2 if world > 1 else 0) if args.permute < 0 else args.permute
    permute = {0: False, 1: True, 2: "owner-sorted"}[pm]
    t0 = time.time()
    n, Ap, Aj = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda", permute=permute, owners=args.owners or world)
    nnz = int(Ap[-1])
    stats = graphgen.degree_stats(Ap)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(args.seed + 300)  # every rank draws the same global values
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    if rank == 0:
        _log(f"[bench] spmv-fp32: R-MAT scale {scale} permute={permute}: n={n} nnz={nnz} deg min/avg/sd/max={stats['min']:.0f}/{stats['avg']:.2f}/{stats['sd']:.1f}/"
             f"{stats['max']:.0f} empty_rows={stats['empty_rows']} gen {time.time() - t0:.1f}s")
    label = {False: "raw", True: "randomly permuted", "owner-sorted": "owner-sorted"}[permute]
    if world > 1:
        return _run_sharded(args, rank, world, scale, label, n, nnz, Ap, Aj, Ax, v)

    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    plan = kernels.CsrPlan(n, nnz, Ap)
    info = plan.attach_cob(n, Ap, Aj, Ax)
    torch.cuda.synchronize()
    attach_ms = (time.perf_counter() - t0) * 1e3
    _log(f"[bench] plan + column-ordered blocks copy built in {attach_ms:.1f} ms: {info}")
    step = lambda: kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0)
    wall, per = time_calls(step, args.steps, args.warmup)
    ms_step = wall / args.steps * 1e3
    k_ms = float(np.mean(per))
    alg = alg_bytes(nnz, n, n)
    gbs = alg / (k_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": kernels.mxv_variant("FLOAT", "MULT", "PLUS", False, plan), "kernel_avg_us": round(k_ms * 1e3, 2), "kernel_median_us": round(float(np.median(per)) * 1e3, 2),
            "algorithmic_bytes": alg, "bytes_per_entry": 8,
            "byte_model": "B = 8*nnz + 4*(n_rows+1) + 4*n_rows (mask) + 4*n_rows (r) + 4*n_cols (v once)   (SURVEY.md 8d; the reference's CSR is the yardstick)"}
    rs = read_stream_gbs()
    roof["measured_read_stream_gbs"], roof["frac_of_measured_read_stream"] = round(rs, 1), round(gbs / rs, 4)
    cg = copy_ceiling_gbs()
    roof["measured_copy_gbs"], roof["frac_of_measured_copy"] = round(cg, 1), round(gbs / cg, 4)
    roof["measured_denominators"] = "read stream: this library's 16-B-per-lane non-temporal read of 1 GiB (the yardstick for a read-dominated kernel); copy: torch's device-to-device copy of 1 GiB, read + write bytes (lower: kept for continuity with rounds 3 - 5)"
    per_call, fname = pmc_traffic(f"spmv_fp32_scale{scale}")
    if per_call and per_call.get("nnz") == nnz:
        roof["traffic"] = per_call["traffic_bytes"]
        roof["traffic_over_algorithmic"] = round(per_call["traffic_bytes"] / alg, 3)
        roof["l1_to_l2_read_requests"] = per_call.get("tcp_tcc_read_req")
        roof["traffic_source"] = f"profiles/{fname}: (2*FETCH_SIZE + WRITE_SIZE)*1024 per call, separate rocprofv3 --pmc passes of this command"
    out = {"metric": "masked SpMV GTEPS", "workload": "spmv-fp32 (north-star configuration; rounds 1-2 quoted the BFS step of configs[1] under this metric: it rides along as variants[0])",
           "value": round(nnz / (wall / args.steps) / 1e9, 3), "unit": "GTEPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"exec_mxv_masked fp32 MULT/PLUS over all rows (op_select ALWAYS, init 0), values and v uniform in [0, 1): R-MAT scale {scale} (ef {args.edge_factor}, {label} labels) "
                                  f"— the north-star ~100 M-nnz scale-free CSR", "n": n, "nnz": nnz, "nnz_selected": nnz, "rows_selected": n,
                      "parallelism": "1 GPU", "layout": info, "layout_build_ms": round(attach_ms, 1),
                      "device_bytes": {"csr": 8 * nnz + 4 * (n + 1), **plan.footprint()}},
           "roofline": roof}
    step()
    torch.cuda.synchronize()
    if not args.no_verify:
        # full-size parity, every row: against the oracle (the reference's sequential fp32 fold) at north_star's 1e-5 relative plus the oracle's own distance
        # from an fp64 fold of the same products on that row (a hub row of 10^5 entries folded one by one in fp32 is ~1e-4 away from its sum: no reordering
        # can agree with it more closely), and against that fp64 fold at 2e-6
        orc = pyoracle.Oracle()
        want = orc.mxv_masked("FLOAT", n, n, Ap.cpu().numpy(), Aj.cpu().numpy(), Ax.cpu().numpy(), mask.cpu().numpy(), v.cpu().numpy(), "MULT", "PLUS", "ALWAYS", 0.0).astype(np.float64)
        got = r.cpu().numpy().astype(np.float64)
        exact = torch.zeros(n, dtype=torch.float64, device="cuda")
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
        exact.index_add_(0, rows, Ax.to(torch.float64) * v[Aj.to(torch.int64)].to(torch.float64))
        ex = exact.cpu().numpy()
        del rows, exact
        den = np.maximum(np.abs(ex), 1e-30)
        err64, ref64 = float(np.max(np.abs(got - ex) / den)), float(np.max(np.abs(want - ex) / den))
        diff = np.abs(got - want)
        plain = diff <= 1e-5 * np.abs(want)
        ok = bool(np.all(diff <= 1e-5 * np.abs(want) + 1.01 * np.abs(want - ex) + 1e-30)) and err64 <= 2e-6
        _log(f"[bench] full-size parity ({n} rows): within 1e-5 of the oracle on {100.0 * plain.mean():.4f} % of the rows, everywhere within 1e-5 + the oracle's own distance from the "
             f"fp64 sum: {ok}; max rel vs the fp64 fold: HIP {err64:.2e}, the reference's sequential fold {ref64:.2e}")
        if not ok:
            raise SystemExit("bench: HIP result differs from the oracle")
        out["verified"] = True
        out["parity"] = {"rows_within_1e-5_of_oracle": float(plain.mean()), "all_rows_within_1e-5_plus_oracle_rounding": ok,
                         "max_rel_vs_fp64": {"hip": err64, "reference_sequential_fold": ref64}}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_reference_sample("FLOAT", Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0, args.cpu_sample_nnz, 1e-5)
    if not args.no_variants:
        out["variants_same_matrix"] = same_matrix_variants(args, n, nnz, Ap, Aj, Ax, v, mask, r)
        out["variants_masked"] = masked_variants(args, plan, n, nnz, Ap, Aj, Ax, v, k_ms)
    plan.close()
    del plan, Ap, Aj, Ax, v, mask, r
    torch.cuda.empty_cache()
    if not args.no_variants and not getattr(args, "scale_given", False):
        t_budget = time.time()
        out["variants"] = []
        out["variants"].append(permuted_headline(args))
        out["variants"] += config_variants(args, t_budget)
    return out


def masked_variants(args, plan, n, nnz, Ap, Aj, Ax, v, all_rows_ms):
    """The same call under NQZERO masks (the reference skips unselected rows, src/cpu/cpu_mxv.hpp:89-92; its own perf test selects every 10th row,
    tests/test_mxv.cpp:91-130): 10 % / 50 % of the rows at random, and the same with every hub row unselected.  GTEPS and roofline fraction are against the SELECTED
    entries (8 nnz_sel + ...); each result is checked against the oracle on a sample of rows."""
    from oracle import pyoracle
    from spla_amd import kernels
    res = []
    steps = max(10, args.steps // 2)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(args.seed + 900)
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    thr = max(16, 2 * nnz // n)
    u = torch.rand(n, generator=gen, device="cuda")
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    orc = pyoracle.Oracle()
    for name, sel in (("10 % of the rows, random", u < 0.1), ("50 % of the rows, random", u < 0.5),
                      (f"10 % of the rows, no hub row (degree < {thr})", (u < 0.1) & (deg < thr)), (f"50 % of the rows, no hub row (degree < {thr})", (u < 0.5) & (deg < thr))):
        m = sel.to(torch.float32)
        nsel = int(deg[sel].sum())
        _, per = time_calls(lambda: kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, m, v, r, "MULT", "PLUS", "NQZERO", 0.0), steps, 3)
        ms = float(np.mean(per))
        alg = alg_bytes(nsel, n, n)
        probe = plan.mask_probe()
        # parity on every 64th row: within 1e-5 of the oracle plus the oracle's own distance from the fp64 fold (the selected-rows kernels fold like the oracle: exact)
        rows, Ap_s, Aj_s, Ax_s = sample_rows(Ap, Aj, Ax, 64)
        mm, vh = int(rows.numel()), v.cpu().numpy()
        want = orc.mxv_masked("FLOAT", mm, n, Ap_s, Aj_s, Ax_s, m[rows].cpu().numpy(), vh, "MULT", "PLUS", "NQZERO", 0.0).astype(np.float64)
        ex = np.bincount(np.repeat(np.arange(mm), np.diff(Ap_s.astype(np.int64))), weights=Ax_s.astype(np.float64) * vh[Aj_s.astype(np.int64)].astype(np.float64), minlength=mm)
        ex = np.where(m[rows].cpu().numpy() != 0, ex, 0.0)
        got = r[rows].cpu().numpy().astype(np.float64)
        ok = bool(np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1.01 * np.abs(want - ex) + 1e-30))
        if not ok:
            raise SystemExit(f"bench: masked variant '{name}' differs from the oracle")
        res.append({"name": "fp32 MULT/PLUS, NQZERO mask: " + name, "rows_selected": int(sel.sum()), "nnz_selected": nsel, "kernel_avg_us": round(ms * 1e3, 2),
                    "ratio_to_all_rows": round(ms / all_rows_ms, 3), "gteps_selected": round(nsel / (ms * 1e-3) / 1e9, 2), "algorithmic_bytes_selected": alg,
                    "roofline_frac_selected": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "served_by": "selected-rows kernels (mxv_rows.hpp)" if probe["served_by_rows"] else "column-ordered blocks (streams the whole copy)",
                    "probe_selected_entries": probe["selected_entries"], "verified_rows": mm})
    return res


def permuted_headline(args):
    """The headline call on the same graph with RANDOM vertex labels (a real graph's labels carry no locality; the generator's put the hubs at the low ids)."""
    from spla_amd import graphgen, kernels
    scale = 23
    n, Ap, Aj = graphgen.rmat_csr(scale, args.edge_factor, seed=args.seed, device="cuda", permute=True)
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(args.seed + 300)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    plan = kernels.CsrPlan(n, nnz, Ap)
    info = plan.attach_cob(n, Ap, Aj, Ax)
    _, per = time_calls(lambda: kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0), max(10, args.steps // 2), 3)
    ms = float(np.mean(per))
    alg = alg_bytes(nnz, n, n)
    exact = torch.zeros(n, dtype=torch.float64, device="cuda")
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
    exact.index_add_(0, rows, Ax.to(torch.float64) * v[Aj.to(torch.int64)].to(torch.float64))
    err = float(((r.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max())
    plan.close()
    return {"name": f"headline call, RANDOM vertex labels (R-MAT scale {scale}, --permute 1)", "n": n, "nnz": nnz, "kernel_avg_us": round(ms * 1e3, 2), "gteps": round(nnz / (ms * 1e-3) / 1e9, 2),
            "roofline": {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes": alg}, "max_rel_vs_fp64": err, "layout": info}


def config_variants(args, t_start):
    """The other BASELINE configs at N = 1, each with its own roofline / cpu_baseline objects (bench_workloads.py): configs[2] the PageRank-step vxm on the com-Orkut
    stand-in, configs[3] the min-plus SSSP step on R-MAT 24, configs[4] mxmT(L, L, L) on R-MAT 20 and mxm on R-MAT 18; plus the BFS step of configs[1]
    (the contract line of rounds 1 - 2).  A workload is skipped once the whole run has used its time budget."""
    import copy
    import bench_workloads
    res = []
    budget_s = float(os.environ.get("SPLA_BENCH_BUDGET_S", 420))
    for key, label in (("pagerank-vxm", "configs[2]"), ("sssp-step", "configs[3] at N = 1"), ("tc-mxmT", "configs[4]: mxmT(L, L, L), R-MAT 20"), ("mxm", "configs[4]: mxm A x A, R-MAT 18"),
                       ("mxm-19", "configs[4]: mxm A x A, R-MAT 19 (3.53 G result entries: the largest the reference's 32-bit row pointers index)")):
        if time.time() - t_start > budget_s:
            res.append({"name": f"{label} ({key})", "skipped": "time budget of the default run used up: python bench.py --workload " + key})
            continue
        b = copy.copy(args)
        b.scale, b.steps, b.warmup, b.workload = 22, 10, 2, key  # (22: bench_workloads maps it to each workload's own size)
        t0 = time.time()
        try:
            line = bench_workloads.WORKLOADS[key](b)
        except Exception as exc:  # a variant must not take the contract line down with it
            res.append({"name": f"{label} ({key})", "error": repr(exc)[:300]})
            torch.cuda.empty_cache()
            continue
        line["name"] = f"BASELINE {label} ({key})"
        line["variant_wall_s"] = round(time.time() - t0, 1)
        res.append(line)
        torch.cuda.empty_cache()
    return res


def same_matrix_variants(args, n, nnz, Ap, Aj, Ax, v, mask, r_cob):
    """Other forms / semirings on the north-star matrix (kernel-only, inputs resident)."""
    from spla_amd import kernels
    res = []
    steps = max(10, args.steps // 2)
    alg = alg_bytes(nnz, n, n)

    def run(name, plan, t, X, m, vec, r, mul, add, sel, init, extra=None):
        _, per = time_calls(lambda: kernels.mxv_masked(plan, t, n, Ap, Aj, X, m, vec, r, mul, add, sel, init), steps, 3)
        ms = float(np.mean(per))
        e = {"name": name, "kernel": kernels.mxv_variant(t, mul, add, False, plan), "kernel_avg_us": round(ms * 1e3, 2), "gteps": round(nnz / (ms * 1e-3) / 1e9, 2),
             "hbm_gbs": round(alg / (ms * 1e-3) / 1e9, 1), "roofline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if extra:
            e.update(extra)
        res.append(e)

    r = torch.empty_like(r_cob)
    csr = kernels.CsrPlan(n, nnz, Ap)
    run("fp32 MULT/PLUS, CSR wave-tile kernel (no layout copy; deterministic fold)", csr, "FLOAT", Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0)
    res[-1]["max_rel_vs_cob"] = float(((r - r_cob).abs() / r_cob.abs().clamp_min(1e-30)).max())
    # ---- the exact-order paths: every row a strict left fold from init in entry order, the reference CPU backend's own loop (src/cpu/cpu_mxv.hpp:94-99), each checked BIT
    # FOR BIT against the oracle on every 64th row and the 4096 longest rows
    from oracle import pyoracle
    from spla_amd import _lib
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    rows = torch.unique(torch.cat([torch.arange(0, n, 64, device="cuda"), torch.topk(deg, 4096).indices]))
    Ap64 = Ap.to(torch.int64)
    dg = deg[rows]
    Aps = torch.zeros(rows.numel() + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(dg, 0, out=Aps[1:])
    rep = torch.repeat_interleave(torch.arange(rows.numel(), device="cuda"), dg)
    src = Ap64[rows][rep] + (torch.arange(int(Aps[-1]), device="cuda") - Aps[:-1][rep])
    sample = (int(rows.numel()), n, Aps.to(torch.int32).cpu().numpy().view(np.uint32), Aj[src].cpu().numpy().view(np.uint32), Ax[src].cpu().numpy(), mask[rows].cpu().numpy(), v.cpu().numpy())
    del rep, src, Aps, dg, deg, Ap64
    orc = pyoracle.Oracle()

    def check_exact(got, mul, add, init):
        want = orc.mxv_masked("FLOAT", *sample, mul, add, "ALWAYS", init)
        g = got[rows].cpu().numpy()
        same = bool(np.all((g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))))
        res[-1]["verified_rows"], res[-1]["bit_identical_to_oracle"] = int(rows.numel()), same
        if not same:
            raise SystemExit(f"bench: {res[-1]['name']} differs from the oracle")

    # (1) the reference-order mode (splahip_set_reference_order; Descriptor::set_reference_order): an fp32 PLUS folded like the reference, the blocks standing aside
    assert _lib.call("splahip_set_reference_order", 1) == 0
    try:
        rr = torch.empty_like(r_cob)
        run("fp32 MULT/PLUS, reference-order mode (every row folded strictly left to right: bit-identical to the reference CPU backend)", csr, "FLOAT", Ax, mask, v, rr,
            "MULT", "PLUS", "ALWAYS", 0.0)
    finally:
        assert _lib.call("splahip_set_reference_order", 0) == 1
    check_exact(rr, "MULT", "PLUS", 0.0)
    res[-1]["max_rel_vs_cob"] = float(((rr - r_cob).abs() / r_cob.abs().clamp_min(1e-30)).max())
    # (2) an op_add that must not be re-ordered (MINUS: neither associative nor commutative): the same ordered kernels with the runtime-switched functor (round 6; a
    # lane per row straight from memory until then: a hub row of 10^5 entries was one lane's)
    run("fp32 MULT/MINUS (an op_add that must not be re-ordered), ordered kernels", csr, "FLOAT", Ax, mask, v, rr, "MULT", "MINUS", "ALWAYS", 0.0)
    check_exact(rr, "MULT", "MINUS", 0.0)
    # (3) a user-defined op_add (the reference's OpBinary::make_float text, compiled with hiprtc): a lane per row up to 508 entries, a wavefront per longer row
    if _lib.call("splahip_jit_available"):
        import ctypes as C
        from spla_amd.ops import BINOP, SELOP
        body = b"(float a, float b) { return a + b; }"  # (PLUS written by the user: the oracle's PLUS is its known answer)
        P_ = lambda t: C.c_void_p(t.data_ptr())
        call = lambda: _lib.call("splahip_mxv_masked_jit", 2, n, P_(Ap), P_(Aj), P_(Ax), P_(mask), P_(v), P_(rr), BINOP["MULT"], None, BINOP["PLUS"], body, SELOP["ALWAYS"], None,
                                 C.c_uint32(0), 0, kernels.current_stream_ptr())
        call()
        torch.cuda.synchronize()
        _, per = time_calls(call, max(3, steps // 4), 1)
        ms = float(np.mean(per))
        res.append({"name": "fp32 MULT / user-defined op_add \"(float a, float b) { return a + b; }\" (hiprtc), exact order for any op", "kernel": "splahip_mxv_jit + splahip_mxv_long_jit",
                    "kernel_avg_us": round(ms * 1e3, 2), "gteps": round(nnz / (ms * 1e-3) / 1e9, 2), "hbm_gbs": round(alg / (ms * 1e-3) / 1e9, 1),
                    "roofline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        check_exact(rr, "MULT", "PLUS", 0.0)
    del rows, rr
    # the reference's own SSSP semiring: fp32 min-plus (src/algorithm.cpp:208-213), 1/8 of v finite, init = +inf
    gen = torch.Generator(device="cuda")
    gen.manual_seed(args.seed + 77)
    w = torch.randint(1, 256, (nnz,), generator=gen, device="cuda").to(torch.float32)
    d = torch.full((n,), float("inf"), dtype=torch.float32, device="cuda")
    d[torch.randint(0, n, (n // 8,), generator=gen, device="cuda")] = 1000.0
    r0 = torch.empty_like(r)
    run("fp32 PLUS/MIN (the reference's SSSP semiring), 1/8 of v finite, CSR kernels", csr, "FLOAT", w, mask, d, r0, "PLUS", "MIN", "ALWAYS", float("inf"))
    cobw = kernels.CsrPlan(n, nnz, Ap)
    cobw.attach_cob(n, Ap, Aj, w)
    r1 = torch.empty_like(r)
    run("fp32 PLUS/MIN (the reference's SSSP semiring), 1/8 of v finite, column-ordered blocks (order-preserving integer keys in LDS)", cobw, "FLOAT", w, mask, d, r1, "PLUS", "MIN",
        "ALWAYS", float("inf"))
    res[-1]["identical_to_csr_kernel"] = bool(torch.equal(r0.view(torch.int32), r1.view(torch.int32)))
    wi = w.to(torch.int32)
    di = torch.where(torch.isinf(d), torch.full_like(d, float(1 << 30)), d).to(torch.int32)
    cobi = kernels.CsrPlan(n, nnz, Ap)
    cobi.attach_cob(n, Ap, Aj, wi)
    mi = torch.zeros(n, dtype=torch.int32, device="cuda")
    ri0, ri1 = torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
    kernels.mxv_masked(csr, "INT", n, Ap, Aj, wi, mi, di, ri0, "PLUS", "MIN", "ALWAYS", 1 << 30)
    run("int32 PLUS/MIN (SSSP step), 1/8 of v finite, column-ordered blocks", cobi, "INT", wi, mi, di, ri1, "PLUS", "MIN", "ALWAYS", 1 << 30)
    res[-1]["identical_to_csr_kernel"] = bool(torch.equal(ri0, ri1))
    for p in (csr, cobw, cobi):
        p.close()
    return res


def _run_sharded(args, rank, world, scale, label, n, nnz, Ap, Aj, Ax, v):
    """N > 1: the product path of the dist layer (bench_dist.ShardedProduct -> splahip_dist_mxv_masked)."""
    import bench_dist
    from oracle import pyoracle
    from spla_amd import sharding
    bounds = sharding.row_splits(Ap, world)
    blk = sharding.RowBlock(Ap, Aj, Ax, bounds, rank)
    del Ap, Aj, Ax
    torch.cuda.empty_cache()
    direct = os.environ.get("SPLA_ALLGATHER", "nccl") == "direct"
    sp = bench_dist.ShardedProduct(blk, blk.Ax, "FLOAT", "MULT", "PLUS", "ALWAYS", 0.0, 0.0, world, rank, direct=direct)
    v_loc = blk.pad_local(v[blk.lo:blk.hi], fill=0.0)
    mask = torch.zeros(blk.n_local, dtype=torch.float32, device="cuda")
    r = torch.empty(blk.n_local, dtype=torch.float32, device="cuda")
    wall, per = bench_dist.timed(lambda: sp.step(mask, v_loc, r), args.steps, args.warmup, world)
    ms = wall / args.steps * 1e3
    phases = bench_dist.phase_report(sp, mask, v_loc, r, max(5, args.steps // 2), world)
    sp.step(mask, v_loc, r)
    torch.cuda.synchronize()
    ok = True
    if not args.no_verify:  # every 64th local row against the oracle on the gathered vector, 1e-5 relative
        orc = pyoracle.Oracle()
        rows, Ap_s, Aj_s, Ax_s = sample_rows(blk.Ap, blk.Aj, blk.Ax, 64)
        m, vf = int(rows.numel()), sp.v_full.cpu().numpy()
        want = orc.mxv_masked("FLOAT", m, blk.n_cols_padded, Ap_s, Aj_s, Ax_s, np.zeros(m, np.float32), vf, "MULT", "PLUS", "ALWAYS", 0.0).astype(np.float64)
        ex = np.bincount(np.repeat(np.arange(m), np.diff(Ap_s.astype(np.int64))), weights=Ax_s.astype(np.float64) * vf[Aj_s.astype(np.int64)].astype(np.float64), minlength=m)
        got = r[rows].cpu().numpy().astype(np.float64)  # 1e-5 of the oracle + the oracle's own distance from the fp64 sum (run(): hub rows)
        ok = bool(np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1.01 * np.abs(want - ex) + 1e-30))
        flag = torch.tensor([1 if ok else 0], device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(int(flag[0]))
        if not ok:
            raise SystemExit("bench: sharded HIP result differs from the oracle")
    k_us = phases.get("local_part_avg_us", 0.0) + phases.get("remote_part_avg_us", 0.0) if sp.two_part else phases.get("whole_part_avg_us", phases["step_avg_us"] - phases["exchange_avg_us"])
    alg = alg_bytes(blk.nnz_local, blk.n_local, blk.n_cols_padded)
    gbs = alg / (k_us * 1e-6) / 1e9 if k_us > 0 else 0.0
    slice_bytes = 4 * blk.max_rows
    line = {"metric": "masked SpMV GTEPS", "value": round(nnz / (ms * 1e-3) / 1e9, 3), "unit": "GTEPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"exec_mxv_masked fp32 MULT/PLUS over all rows (op_select ALWAYS, init 0): R-MAT scale {scale} = 23 + log2({world}) (ef {args.edge_factor}, {label} labels), "
                                   f"1-D nnz-balanced row blocks x{world}, all-gather of the dense fp32 vector per step", "n": n, "nnz": nnz, "nnz_rank0": blk.nnz_local,
                       "rows_rank0": blk.n_local, "parallelism": f"1-D row blocks x{world} + RCCL all-gather(v) overlapped with the local-column part (splahip_dist_mxv_masked)"},
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "mxv_cob (local-column part) + mxv_cob accumulate (remote-column part)" if sp.two_part else "unsplit block", "kernel_avg_us": round(k_us, 2),
                         "algorithmic_bytes": alg, "note": "rank 0's row block; the parts timed on their own"},
            "exchange": {"mode": ("C ABI: splahip_dist_mxv_masked, RCCL " + ("grouped send/recv per peer (direct)" if sp.direct else "ncclAllGather") + " on the communicator's stream")
                                 if sp.rccl else "torch.distributed (" + dist.get_backend() + " debug backend) between the C-ABI phases",
                         "bytes_per_peer": slice_bytes, "bytes_received_per_rank": slice_bytes * (world - 1), "two_part_form": sp.two_part,
                         "gbs_per_link_if_direct": round(slice_bytes / (phases["exchange_avg_us"] * 1e-6) / 1e9, 2) if phases["exchange_avg_us"] else None, **phases},
            "shard": {"nnz_local_columns": sp.shard.nnz_local_cols, "nnz_remote_columns": sp.shard.nnz_remote_cols}, "verified": ok}
    if sp.fallback:
        line["exchange"]["fallback"] = sp.fallback
    sp.close()
    del sp, blk, r, mask, v_loc
    torch.cuda.empty_cache()
    if not args.no_variants:  # BASELINE configs[3] in the same command: the STRONG split of the min-plus SSSP step on R-MAT scale 24 over the same dist layer
        import copy
        b = copy.copy(args)
        b.scale = 22 if not getattr(args, "scale_given", False) else args.scale + 1  # (bench_dist maps 22 to configs[3]'s own scale, 24; a reduced debug run scales it down too)
        v = bench_dist.sssp_step_sharded(b, rank, world)
        v["name"] = "SSSP step (BASELINE configs[3]): min-plus int32 on R-MAT scale 24, strong 1-D row split, all-gather of the distance vector per step"
        line["variants"] = [v]
        budget = float(os.environ.get("SPLA_BENCH_BUDGET_S", "420"))
        # (both extra variants only while the command is still young: the graph of the headline is built once more; _T0 = import time of this module)
        def still_young(limit):  # one answer for every rank (the variants run collectives): the slowest rank's clock decides
            flag = torch.tensor([1 if time.time() - _T0 < limit else 0], device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag[0]))

        if args.permute < 0 and still_young(budget):  # the headline ran on owner-sorted labels (the default for N > 1): the same step on the generator's own labels
            c = copy.copy(args)
            c.permute, c.no_variants, c.no_cpu_baseline = 0, True, True
            raw = run(c, rank, world)
            line["variants"].append({"name": "headline step, RAW (generator) vertex labels (--permute 0): hub rows and hub columns sit in the first ranks' blocks",
                                     **{k: raw[k] for k in ("value", "unit", "ms_per_step", "roofline", "exchange", "shard", "verified") if k in raw}})
        if still_young(1.5 * budget):  # BASELINE configs[4] across the ranks: the triangle-count call by rows of the mask, B whole on every rank
            c = copy.copy(args)
            c.workload, c.scale = "tc-mxmT", (22 if not getattr(args, "scale_given", False) else max(10, args.scale - 3))
            t = bench_dist.spgemm_sharded(c, rank, world)
            t["name"] = "BASELINE configs[4] across the ranks: mxmT(L, L, L) by rows of the mask (strong scaling; no exchange in the data path)"
            line["variants"].append(t)
    return line
