"""Round 5 A/B: the static pieces (SPLAHIP_COB_DYN=0) against the run-time schedule (mxv_cob_dyn.hpp) on the north-star call and the min-plus calls.
    python tools/exp/exp_r5_dyn.py 23 [KEY=VAL ...]   (extra knobs apply to both sides)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402
from tools.exp.exp_r4_cob import timeit  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
for kv in sys.argv[2:]:
    k, v_ = kv.split("=", 1)
    os.environ[k] = v_
_lib.load()
_lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
gen = torch.Generator(device="cuda")
gen.manual_seed(101)
Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
zero = torch.zeros(n, dtype=torch.float32, device="cuda")
alg = 8 * nnz + 4 * (n + 1) + 8 * n + 4 * n
exact = torch.zeros(n, dtype=torch.float64, device="cuda")
rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
exact.index_add_(0, rows, Ax.to(torch.float64) * v[Aj.to(torch.int64)].to(torch.float64))
del rows
w = torch.randint(1, 256, (nnz,), generator=gen, device="cuda").to(torch.int32)
d = torch.full((n,), 1 << 30, dtype=torch.int32, device="cuda")
d[torch.randint(0, n, (n // 8,), generator=gen, device="cuda")] = 1000
zi = torch.zeros(n, dtype=torch.int32, device="cuda")
res = {}
for rep in range(2):
    for dyn in ("0", "1"):
        os.environ["SPLAHIP_COB_DYN"] = dyn
        _lib.env_refresh()
        r = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
        p = kernels.CsrPlan(n, nnz, Ap)
        p.attach_cob(n, Ap, Aj, Ax)
        a, amin = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, zero, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
        rel = ((r.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
        p.close()
        ri = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        p = kernels.CsrPlan(n, nnz, Ap)
        p.attach_cob(n, Ap, Aj, w)
        c, cmin = timeit(lambda: kernels.mxv_masked(p, "INT", n, Ap, Aj, w, zi, d, ri, "PLUS", "MIN", "ALWAYS", 1 << 30))
        p.close()
        same = ""
        if dyn == "0":
            res["int"] = ri.clone()
        else:
            same = f"int result equal to the static pieces': {bool(torch.equal(res['int'], ri))}"
        print(f"scale={scale} dyn={dyn}: fp32 plus-times {a:7.1f} us (min {amin:7.1f}) {100 * alg / a / 1e6 / 8:5.1f} % of 8 TB/s, max rel vs fp64 {rel:.2e};  int32 min-plus {c:7.1f} us (min {cmin:7.1f})  {same}", flush=True)
