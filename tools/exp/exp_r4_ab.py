"""Round 4 A/B: the north-star call and the min-plus calls over the column-ordered blocks at R-MAT `scale`, for whichever library SPLA_LIB_TAG names."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402
from tools.exp.exp_r4_cob import timeit  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
_lib.load()
_lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
gen = torch.Generator(device="cuda")
gen.manual_seed(101)
Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
zero = torch.zeros(n, dtype=torch.float32, device="cuda")
r = torch.empty(n, dtype=torch.float32, device="cuda")
p = kernels.CsrPlan(n, nnz, Ap)
p.attach_cob(n, Ap, Aj, Ax)
a, _ = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, zero, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
p.close()
w = torch.randint(1, 256, (nnz,), generator=gen, device="cuda").to(torch.float32)
d = torch.full((n,), float("inf"), dtype=torch.float32, device="cuda")
d[torch.randint(0, n, (n // 8,), generator=gen, device="cuda")] = 1000.0
p = kernels.CsrPlan(n, nnz, Ap)
p.attach_cob(n, Ap, Aj, w)
b, _ = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, w, zero, d, r, "PLUS", "MIN", "ALWAYS", float("inf")))
p.close()
wi = w.to(torch.int32)
di = torch.where(torch.isinf(d), torch.full_like(d, float(1 << 30)), d).to(torch.int32)
zi = torch.zeros(n, dtype=torch.int32, device="cuda")
ri = torch.empty(n, dtype=torch.int32, device="cuda")
p = kernels.CsrPlan(n, nnz, Ap)
p.attach_cob(n, Ap, Aj, wi)
c, _ = timeit(lambda: kernels.mxv_masked(p, "INT", n, Ap, Aj, wi, zi, di, ri, "PLUS", "MIN", "ALWAYS", 1 << 30))
p.close()
print(f"tag={os.environ.get('SPLA_LIB_TAG', '-')} scale={scale}: fp32 plus-times {a:.1f} us   fp32 min-plus {b:.1f} us   int32 min-plus {c:.1f} us", flush=True)
