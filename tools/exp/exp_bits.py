"""BFS-step timing of the sliced-ELL BITS form on R-MAT scale 22 under a few mask / vector settings (which term dominates?)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from spla_amd import _lib, graphgen, kernels
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(22, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
Ax = torch.ones(nnz, dtype=torch.int32, device="cuda")
# level-2 state of a BFS from vertex 0: frontier = out-neighbours of 0, visited = {0} + frontier
front = Aj[int(Ap[0]):int(Ap[1])].to(torch.int64)
v = torch.zeros(n, dtype=torch.int32, device="cuda"); v[front] = 1
depth = torch.zeros(n, dtype=torch.int32, device="cuda"); depth[front] = 2; depth[0] = 1
zeros = torch.zeros(n, dtype=torch.int32, device="cuda")
deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
print("frontier", int(v.sum()), "selected nnz", int(deg[depth == 0].sum()), "of", nnz)
sell = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax)
r = torch.empty(n, dtype=torch.int32, device="cuda")
def t(name, mask, vec, sel, form):
    os.environ["SPLAHIP_SELL_BITS"] = form
    fn = lambda: kernels.mxv_masked(sell, "INT", n, Ap, Aj, Ax, mask, vec, r, "BAND", "BOR", sel, 0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{sys.argv[1] if len(sys.argv) > 1 else '':8s} form={form} {name:44s} {a.elapsed_time(b) / 20 * 1e3:8.1f} us", flush=True)
only = os.environ.get("SPLA_EXP_ONLY")
for form in ("2", "0"):
    if only == "A":
        t("A bfs step (depth EQZERO, frontier)", depth, v, "EQZERO", form)
        continue
    if only == "B":
        t("B all rows ALWAYS, frontier", depth, v, "ALWAYS", form)
        continue
    t("A bfs step (depth EQZERO, frontier)", depth, v, "EQZERO", form)
    t("B all rows ALWAYS, frontier", depth, v, "ALWAYS", form)
    t("C all rows via EQZERO on zero mask, frontier", zeros, v, "EQZERO", form)
    t("D bfs mask, v all zero", depth, zeros, "EQZERO", form)
    t("E nothing selected (NEVER)", depth, v, "NEVER", form)
