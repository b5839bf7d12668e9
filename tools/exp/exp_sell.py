"""A/B: CSR wave-tile kernel vs the sliced-ELL + LDS-hot-column kernel on R-MAT (fp32 plus-times all rows; int32 min-plus; masked)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from spla_amd import _lib, graphgen, kernels

torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
g = torch.Generator(device="cuda"); g.manual_seed(3)
Ax = torch.rand(nnz, generator=g, device="cuda")
v = torch.rand(n, generator=g, device="cuda")
mask = torch.zeros(n, device="cuda")
t0 = time.time()
csr = kernels.CsrPlan(n, nnz, Ap)
torch.cuda.synchronize(); t1 = time.time()
sell = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax)
torch.cuda.synchronize(); t2 = time.time()
print(f"n={n} nnz={nnz} csr plan {1e3*(t1-t0):.1f} ms, sell plan {1e3*(t2-t1):.1f} ms: {sell.sell}", flush=True)

def bench(name, plan, t, ax, vv, mk, om, oa, sel, init, steps=20):
    r = torch.empty(n, dtype=vv.dtype, device="cuda")
    fn = lambda: kernels.mxv_masked(plan, t, n, Ap, Aj, ax, mk, vv, r, om, oa, sel, init)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / steps * 1e3
    print(f"{name:44s} {us:9.1f} us  {nnz / us / 1e3:8.1f} GTEPS(all nnz)  {kernels.mxv_variant(t, om, oa, False, plan)}", flush=True)
    return r.clone()

ra = bench("fp32 plus-times, all rows: CSR", csr, "FLOAT", Ax, v, mask, "MULT", "PLUS", "ALWAYS", 0.0)
rb = bench("fp32 plus-times, all rows: SELL", sell, "FLOAT", Ax, v, mask, "MULT", "PLUS", "ALWAYS", 0.0)
err = ((ra - rb).abs() / ra.abs().clamp_min(1e-30)).max().item()
print("max rel diff CSR vs SELL:", err)
ra = bench("fp32 plus-times, EQZERO(all sel): CSR", csr, "FLOAT", Ax, v, mask, "MULT", "PLUS", "EQZERO", 0.0)
rb = bench("fp32 plus-times, EQZERO(all sel): SELL", sell, "FLOAT", Ax, v, mask, "MULT", "PLUS", "EQZERO", 0.0)
# int32 min-plus
Axi = torch.randint(1, 256, (nnz,), generator=g, device="cuda", dtype=torch.int32)
vi = torch.where(torch.rand(n, generator=g, device="cuda") < 0.125, torch.randint(0, 1000, (n,), generator=g, device="cuda", dtype=torch.int32), torch.full((n,), 1 << 30, device="cuda", dtype=torch.int32))
mi = torch.zeros(n, dtype=torch.int32, device="cuda")
selli = kernels.CsrPlan(n, nnz, Ap, n, Aj, Axi)
ra = bench("int32 min-plus, all rows: CSR", csr, "INT", Axi, vi, mi, "PLUS", "MIN", "ALWAYS", 1 << 30)
rb = bench("int32 min-plus, all rows: SELL", selli, "INT", Axi, vi, mi, "PLUS", "MIN", "ALWAYS", 1 << 30)
print("int min-plus identical:", bool(torch.equal(ra, rb)))
# BFS-like: mask selects rows with depth==0: mark the 3% highest-degree rows visited
deg = (Ap[1:] - Ap[:-1])
thr = torch.sort(deg, descending=True).values[int(0.03 * n)]
depth = (deg > thr).to(torch.int32)
ones = torch.ones(nnz, dtype=torch.int32, device="cuda")
front = (torch.rand(n, generator=g, device="cuda") < 0.02).to(torch.int32)
sellb = kernels.CsrPlan(n, nnz, Ap, n, Aj, ones)
sel_nnz = int(deg[depth == 0].sum())
print("selected nnz fraction", sel_nnz / nnz)
ra = bench("int32 BAND/BOR EQZERO(depth): CSR", csr, "INT", ones, front, depth, "BAND", "BOR", "EQZERO", 0)
rb = bench("int32 BAND/BOR EQZERO(depth): SELL", sellb, "INT", ones, front, depth, "BAND", "BOR", "EQZERO", 0)
print("bfs-like identical:", bool(torch.equal(ra, rb)))
