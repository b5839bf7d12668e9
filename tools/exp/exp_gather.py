"""Experiment: what bounds mxv_rowblock?  Same CSR structure (R-MAT scale 22), fp32 MULT/PLUS, all rows selected,
but the column indices replaced by synthetic patterns that change ONLY the gather behaviour."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from spla_amd import _lib, graphgen, kernels

torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 22
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
plan = kernels.CsrPlan(n, nnz, Ap)
HOT = '--sell' in sys.argv  # run through the sliced-ELL copy (values copied: rebuilt per column pattern)
SEL = 'EQZERO' if '--eqzero' in sys.argv else 'ALWAYS'
g = torch.Generator(device="cuda"); g.manual_seed(3)
Ax = torch.rand(nnz, generator=g, device="cuda")
v = torch.rand(n, generator=g, device="cuda")
mask = torch.zeros(n, device="cuda"); r = torch.empty(n, device="cuda")

def bench(name, aj, steps=20):
    pl = kernels.CsrPlan(n, nnz, Ap, n, aj, Ax) if HOT else plan
    fn = lambda: kernels.mxv_masked(pl, "FLOAT", n, Ap, aj, Ax, mask, v, r, "MULT", "PLUS", SEL, 0.0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / steps * 1e3
    print(f"{name:46s} {us:9.1f} us  {nnz / us / 1e3:8.1f} GTEPS  stream {8 * nnz / us / 1e3:8.1f} GB/s", flush=True)

print(f"n={n} nnz={nnz} blocks={plan.n_blocks}")
k = torch.arange(nnz, device="cuda", dtype=torch.int64)
bench("real R-MAT columns", Aj)
bench("coalesced: k % 64", (k % 64).to(torch.int32))
bench("coalesced: k % n (stream v)", (k % n).to(torch.int32))
for T in [1 << 10, 1 << 13, 1 << 16, 1 << 18, 1 << 20, 1 << 22]:
    if T <= n:
        bench(f"uniform random in [0,{T}) ({T * 4 // 1024} KiB)", torch.randint(0, T, (nnz,), generator=g, device="cuda", dtype=torch.int32))
perm = torch.randperm(n, generator=g, device="cuda")
bench("real columns, randomly relabelled", perm[Aj.to(torch.int64)].to(torch.int32))
# sort columns by in-degree (hub clustering): hottest columns get the smallest ids
indeg = torch.bincount(Aj.to(torch.int64), minlength=n)
order = torch.argsort(indeg, descending=True)
newid = torch.empty_like(order); newid[order] = torch.arange(n, device="cuda")
bench("real columns, relabelled by in-degree rank", newid[Aj.to(torch.int64)].to(torch.int32))
bench("relabelled by in-degree rank after random", newid[Aj.to(torch.int64)].to(torch.int32))
