"""Diagnostic: wall time of the sliced-ELL copy's build (splahip_csr_plan_attach_sell) and of the column-ordered blocks' on R-MAT <scale>; with
SPLAHIP_SELL_TRACE=1 the phases of the former on stderr."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from spla_amd import _lib, graphgen, kernels

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
torch.cuda.set_device(0)
_lib.load()
_lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
Ax = torch.ones(nnz, dtype=torch.int32, device="cuda")
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    info = p.attach_cob(n, Ap, Aj, Ax)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"R-MAT {scale} rep {rep}: plan + sliced-ELL copy {1e3 * (t1 - t0):.1f} ms; column-ordered blocks {1e3 * (t2 - t1):.1f} ms; footprint {p.footprint()}", flush=True)
    p.close()
