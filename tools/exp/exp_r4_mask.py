"""Round 4: masked fp32 plus-times calls (NQZERO) over the blocks + mask probe: random 10 % / 50 % of the rows, and the same with every hub row unselected.
    python tools/exp/exp_r4_mask.py 23 [KEY=VAL ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402
from tools.exp.exp_r4_cob import timeit  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    for kv in sys.argv[2:]:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    _lib.load()
    _lib.call("splahip_init", 0)
    n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(101)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
    p = kernels.CsrPlan(n, nnz, Ap)
    p.attach_cob(n, Ap, Aj, Ax)
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    zero = torch.zeros(n, dtype=torch.float32, device="cuda")
    us_all, _ = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, zero, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
    print(f"R-MAT {scale}: all rows (ALWAYS) {us_all:.1f} us", flush=True)
    thr = max(16, 2 * nnz // n)
    u = torch.rand(n, generator=gen, device="cuda")
    for name, sel in (("random 10 %", u < 0.1), ("random 50 %", u < 0.5), ("10 %, no hub row", (u < 0.1) & (deg < thr)), ("50 %, no hub row", (u < 0.5) & (deg < thr))):
        mask = sel.to(torch.float32)
        nsel = int(deg[sel].sum())
        alg = 8 * nsel + 4 * (n + 1) + 8 * n + 4 * n
        us, mn = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "NQZERO", 0.0))
        pr = p.mask_probe()
        print(f"  {name:18s}: selected nnz {nsel / 1e6:7.2f} M ({nsel / nnz:.3f})  {us:7.1f} us = {us / us_all:.2f} x all rows   {nsel / us / 1e3:6.1f} GTEPS of selected nnz   "
              f"{100 * alg / us / 1e6 / 8:5.1f} % of 8 TB/s   probe {pr}", flush=True)
    p.close()


if __name__ == "__main__":
    main()
