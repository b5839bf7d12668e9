"""Round 6: ONE masked fp32 plus-times call shape (NQZERO) over the blocks + mask probe, for a kernel trace per mask shape.
    python tools/exp/exp_r6_mask.py 23 <random10|random50|light10|light50|stride10> [KEY=VAL ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402
from tools.exp.exp_r4_cob import timeit  # noqa: E402


def main():
    scale, which = int(sys.argv[1]), sys.argv[2]
    for kv in sys.argv[3:]:
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
    thr = max(16, 2 * nnz // n)
    u = torch.rand(n, generator=gen, device="cuda")
    sel = {"random10": u < 0.1, "random50": u < 0.5, "light10": (u < 0.1) & (deg < thr), "light50": (u < 0.5) & (deg < thr),
           "stride10": (torch.arange(n, device="cuda") % 10) == 0}[which]
    mask = sel.to(torch.float32)
    nsel = int(deg[sel].sum())
    us, mn = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "NQZERO", 0.0))
    print(f"R-MAT {scale} {which}: selected rows {int(sel.sum())} nnz {nsel / 1e6:.2f} M ({nsel / nnz:.3f})  {us:.1f} us (min {mn:.1f})  probe {p.mask_probe()}", flush=True)
    p.close()


if __name__ == "__main__":
    main()
