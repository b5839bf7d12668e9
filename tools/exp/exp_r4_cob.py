"""Round 4: time the fp32 plus-times call over the column-ordered blocks (all rows) on R-MAT `scale`, raw or randomly relabelled.
    SPLA_LIB_TAG=<tag> python tools/exp/exp_r4_cob.py 23 [permute:0|1] [env KEY=VAL ...]
Prints one line: tag, scale, permute, us per call (median of 30), % of 8 TB/s on the algorithmic bytes, max rel error vs fp64."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402


def timeit(fn, steps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) * 1e3 for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    permute = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    for kv in sys.argv[3:]:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    _lib.load()
    _lib.call("splahip_init", 0)
    n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda", permute=permute)
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(101)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    alg = 8 * nnz + 4 * (n + 1) + 8 * n + 4 * n
    p = kernels.CsrPlan(n, nnz, Ap)
    info = p.attach_cob(n, Ap, Aj, Ax)
    r = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
    us, mn = timeit(lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
    rel = -1.0
    if os.environ.get("EXP_CHECK", "1") == "1":
        exact = torch.zeros(n, dtype=torch.float64, device="cuda")
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
        exact.index_add_(0, rows, Ax.to(torch.float64) * v[Aj.to(torch.int64)].to(torch.float64))
        rel = ((r.to(torch.float64) - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
    print(f"tag={os.environ.get('SPLA_LIB_TAG', '-')} scale={scale} permute={int(permute)} extra={sys.argv[3:]} : {us:7.1f} us (min {mn:7.1f})  "
          f"{100 * alg / us / 1e6 / 8:5.1f} % of 8 TB/s  max rel vs fp64 {rel:.2e}  {info}", flush=True)
    p.close()


if __name__ == "__main__":
    main()
