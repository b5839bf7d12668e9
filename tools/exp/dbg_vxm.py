import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import bench_workloads as bw
from spla_amd import kernels, _lib
from oracle import pyoracle
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
scale = 20
n, Ap, Aj = bw._sym_rmat(scale, 28, 1)
nnz = int(Ap[-1])
deg = (Ap[1:] - Ap[:-1]).to(torch.float32)
rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
Ax = (0.85 / deg.clamp(min=1.0))[rows].contiguous()
Tp, Tj, Tx = kernels.csr_transpose(n, n, Ap, Aj, Ax)
plan_T = kernels.CsrPlan(n, nnz, Tp)
vi = torch.arange(n, device="cuda", dtype=torch.int32); vx = torch.full((n,), 1.0 / n, device="cuda"); mask = torch.zeros(n, device="cuda")
ri, rx = kernels.vxm_pull(plan_T, "FLOAT", n, n, Tp, Tj, Tx, mask, vi, vx, "MULT", "PLUS", "ALWAYS", identity=True)
orc = pyoracle.Oracle()
wi, wx = orc.vxm_masked("FLOAT", n, n, Ap.cpu().numpy(), Aj.cpu().numpy(), Ax.cpu().numpy(), mask.cpu().numpy(), vi.cpu().numpy(), vx.cpu().numpy(), "MULT", "PLUS", "ALWAYS")
gi, gx = ri.cpu().numpy().view(np.uint32), rx.cpu().numpy()
print("idx equal", np.array_equal(wi, gi), len(wi), len(gi))
if np.array_equal(wi, gi):
    rel = np.abs(gx - wx) / np.abs(wx)
    indeg = np.diff(Tp.cpu().numpy().astype(np.int64))[gi]
    print("max rel", rel.max(), "n > 1e-5:", int((rel > 1e-5).sum()), "indeg of worst", indeg[rel.argmax()], "median indeg of bad", np.median(indeg[rel > 1e-5]) if (rel>1e-5).any() else None)
    # float64 reference
    x64 = np.zeros(n); np.add.at(x64, Aj.cpu().numpy(), (Ax.cpu().numpy().astype(np.float64) * (1.0 / n)))
    print("vs f64: hip max rel", (np.abs(gx - x64[gi]) / x64[gi]).max(), " oracle max rel", (np.abs(wx - x64[gi]) / x64[gi]).max())
