import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from spla_amd import _lib, graphgen, kernels
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(22, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
g = torch.Generator(device="cuda"); g.manual_seed(3)
Ax = torch.rand(nnz, generator=g, device="cuda"); v = torch.rand(n, generator=g, device="cuda"); mask = torch.zeros(n, device="cuda")
indeg = torch.bincount(Aj.to(torch.int64), minlength=n)
top = torch.topk(indeg, 40000).values.sum().item()
print("coverage of the 40000 most referenced columns:", top / nnz, " of 8192:", torch.topk(indeg, 8192).values.sum().item() / nnz)
sell = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax)
r = torch.empty(n, device="cuda")
fn = lambda: kernels.mxv_masked(sell, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0)
for _ in range(3): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fn()
b.record(); torch.cuda.synchronize()
print(sys.argv[1] if len(sys.argv) > 1 else "", f"{a.elapsed_time(b) / 20 * 1e3:.1f} us")
