"""Round 5: the fp32 plus-times call under one schedule, for a kernel trace:  SPLAHIP_COB_DYN=0|1 python tools/exp/exp_r5_dyn_once.py 23"""
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
r = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
p = kernels.CsrPlan(n, nnz, Ap)
p.attach_cob(n, Ap, Aj, Ax)
a, amin = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, zero, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
print(f"scale={scale} dyn={os.environ.get('SPLAHIP_COB_DYN', '-')}: {a:.1f} us (min {amin:.1f})")
