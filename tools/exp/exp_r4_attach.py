"""Round 4: where the attach of the column-ordered blocks spends its time (SPLAHIP_COB_TRACE prints the phases), R-MAT `scale`, twice (the second attach finds warm allocators)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["SPLAHIP_COB_TRACE"] = "1"
from spla_amd import _lib, graphgen, kernels  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
_lib.load()
_lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
Ax = torch.rand(nnz, device="cuda", dtype=torch.float32)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = kernels.CsrPlan(n, nnz, Ap)
    t1 = time.perf_counter()
    p.attach_cob(n, Ap, Aj, Ax)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"== rep {rep}: plan {1e3 * (t1 - t0):.1f} ms, attach {1e3 * (t2 - t1):.1f} ms", flush=True)
    p.close()
