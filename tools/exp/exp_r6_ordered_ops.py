"""Round 6: what the exact-order paths cost on the north-star matrix: a built-in non-commutative op_add (Variant::RowSeq), a user-defined op_add (hiprtc), fp32 MULT as op_add.
    python tools/exp/exp_r6_ordered_ops.py [scale]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402
from spla_amd.ops import BINOP, SELOP  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    _lib.load()
    _lib.call("splahip_init", 0)
    n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
    nnz = int(Ap[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
    v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
    mask = torch.zeros(n, dtype=torch.float32, device="cuda")
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    plan = kernels.CsrPlan(n, nnz, Ap)
    deg = (Ap[1:] - Ap[:-1])
    print(f"R-MAT {scale}: n={n} nnz={nnz} max degree {int(deg.max())}", flush=True)

    def timed(name, fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print(f"  {name:70s} {(time.perf_counter() - t0) / reps * 1e3:10.3f} ms", flush=True)

    timed("built-in MULT/PLUS (wave-tile kernel)", lambda: kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0))
    for add in ("MINUS", "SECOND", "MULT"):
        timed(f"built-in MULT/{add}: {kernels.mxv_variant('FLOAT', 'MULT', add)}", lambda a=add: kernels.mxv_masked(plan, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", a, "ALWAYS", 0.5))
    if _lib.call("splahip_jit_available"):
        src = b"(float a, float b) { return 0.25f * a + 0.75f * b; }"
        ptr = lambda t: C.c_void_p(t.data_ptr())
        def jit():
            _lib.call("splahip_mxv_masked_jit", 2, n, ptr(Ap), ptr(Aj), ptr(Ax), ptr(mask), ptr(v), ptr(r), BINOP["MULT"], None, 0, src, SELOP["ALWAYS"], None,
                      C.c_uint32(0), 0, None)
        timed("user-defined op_add (hiprtc): 0.25 a + 0.75 b", jit, reps=2)
    plan.close()


if __name__ == "__main__":
    main()
