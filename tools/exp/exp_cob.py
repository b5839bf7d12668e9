"""Time the column-ordered-blocks SpMV (fp32 plus-times, all rows) against the sliced-ELL and CSR kernels on R-MAT, over row_bits / tile_chunks."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402


def timeit(fn, steps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    configs = sys.argv[2:] or ["14:256", "14:128", "14:512", "15:256", "13:256"]
    _lib.load()
    _lib.call("splahip_init", 0)
    n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(101)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    alg = 8 * nnz + 4 * (n + 1) + 8 * n + 4 * n
    print(f"R-MAT scale {scale}: n={n} nnz={nnz} algorithmic bytes {alg / 1e6:.0f} MB", flush=True)
    ref = torch.empty(n, dtype=torch.float32, device="cuda")
    plan = kernels.CsrPlan(n, nnz, Ap)
    f = lambda: kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, ref, "MULT", "PLUS", "ALWAYS", 0.0)
    us = timeit(f)
    print(f"csr wave-tile          : {us:8.1f} us  {alg / us / 1e6:6.2f} TB/s  frac {alg / us / 1e6 / 8:.3f}", flush=True)
    exact = torch.zeros(n, dtype=torch.float64, device="cuda")
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
    exact.index_add_(0, rows, Ax.to(torch.float64) * v[Aj.to(torch.int64)].to(torch.float64))
    del rows
    if os.environ.get("EXP_SELL", "1") == "1":
        t0 = time.perf_counter()
        ps = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r = torch.empty_like(ref)
        us = timeit(lambda: kernels.mxv_masked(ps, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
        print(f"sliced-ELL             : {us:8.1f} us  {alg / us / 1e6:6.2f} TB/s  frac {alg / us / 1e6 / 8:.3f}   (attach {1e3 * (t1 - t0):.0f} ms)", flush=True)
        ps.close()
    for cfg in configs:
        parts = [int(x) for x in cfg.split(":")]
        rb, tc = parts[0], parts[1]
        if len(parts) > 2:  # rb:tc:share_pct:line_pct[:hub_ranges] — the schedule knobs of the column-ordered blocks (read at attach time)
            os.environ["SPLAHIP_COB_SHARE_PCT"] = str(parts[2])
            os.environ["SPLAHIP_COB_LINE_PCT"] = str(parts[3])
        if len(parts) > 4:
            os.environ["SPLAHIP_COB_HUB_RANGES"] = str(parts[4])
        if len(parts) > 5:  # ... :natural_slots
            os.environ["SPLAHIP_COB_NAT_SLOTS"] = str(parts[5])
        p = kernels.CsrPlan(n, nnz, Ap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = p.attach_cob(n, Ap, Aj, Ax, rb, tc)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
        us = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
        rel = ((r.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
        relc = ((ref.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
        print(f"cob {cfg}: {us:8.1f} us  {alg / us / 1e6:6.2f} TB/s  frac {alg / us / 1e6 / 8:.3f}   attach {1e3 * (t1 - t0):.0f} ms  {info}  "
              f"max rel vs fp64 {rel:.2e} (csr kernel {relc:.2e})", flush=True)
        if os.environ.get("EXP_INT", "0") == "1":
            w_i = torch.randint(1, 256, (nnz,), generator=gen, device="cuda", dtype=torch.int32)
            d_i = torch.full((n,), 1 << 30, dtype=torch.int32, device="cuda")
            d_i[torch.randint(0, n, (n // 8,), generator=gen, device="cuda")] = 1000
            zm = torch.zeros(n, dtype=torch.int32, device="cuda")
            pi = kernels.CsrPlan(n, nnz, Ap)
            r0 = torch.empty(n, dtype=torch.int32, device="cuda")
            kernels.mxv_masked(pi, "INT", n, Ap, Aj, w_i, zm, d_i, r0, "PLUS", "MIN", "ALWAYS", 1 << 30)
            pi.attach_cob(n, Ap, Aj, w_i, rb, tc)
            r1 = torch.empty_like(r0)
            us = timeit(lambda: kernels.mxv_masked(pi, "INT", n, Ap, Aj, w_i, zm, d_i, r1, "PLUS", "MIN", "ALWAYS", 1 << 30))
            print(f"    int32 min-plus        : {us:8.1f} us  frac {alg / us / 1e6 / 8:.3f}  identical to csr kernel: {bool(torch.equal(r0, r1))}", flush=True)
            pi.close()
        p.close()


if __name__ == "__main__":
    main()
