"""Helpers for the -m gpu parity tests: numpy <-> device tensors and one-call wrappers around the HIP path."""
from __future__ import annotations

import numpy as np
import torch

from spla_amd import kernels


def dev(a: np.ndarray) -> torch.Tensor:
    """numpy array of 4-byte words -> device tensor (uint32 travels as int32 bits)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def host(t: torch.Tensor, np_dtype) -> np.ndarray:
    a = t.detach().cpu().numpy()
    return a.view(np_dtype) if a.dtype != np_dtype else a


def hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, op_mult, op_add, op_select, init, early_exit=False, junk=None):
    """Run the masked SpMV three times — the CSR wave-tile / row kernels (SPLAHIP_SELL=0), the experimental persistent
    LDS-hot-cache CSR kernel (SPLAHIP_HOT=1) and the product path (sliced-ELL copy attached) — and cross-check them: the two CSR
    forms are bit-identical (same work items, same fold order); the sliced-ELL form is bit-identical for int / uint and within
    1e-5 for fp32 (it folds every row <= 256 entries strictly in entry order, the CSR form uses an in-order tree for rows > 32).
    Returns the product path's result."""
    from helpers import NP
    dAp, dAj, dAx, dm, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(mask, NP[t])), dev(np.asarray(v, NP[t]))
    import os
    outs = {}
    for form in ("csr", "hot", "sell"):
        if form == "sell":
            plan = kernels.CsrPlan(n_rows, len(Aj), dAp, n_cols, dAj, dAx)
        else:
            plan = kernels.CsrPlan(n_rows, len(Aj), dAp, n_cols if form == "hot" else 0, dAj if form == "hot" else None)
        r = torch.empty(n_rows, dtype=kernels.torch_dtype(t), device="cuda")
        if junk is not None:
            r.fill_(junk)
        os.environ["SPLAHIP_HOT"] = "1" if form == "hot" else "0"  # the persistent CSR kernel is experimental: switched on per call here
        try:
            kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dm, dv, r, op_mult, op_add, op_select, init, early_exit)
            torch.cuda.synchronize()
        finally:
            os.environ["SPLAHIP_HOT"] = "0"
        outs[form] = host(r, NP[t])
        plan.close()
    assert np.array_equal(outs["csr"].view(np.uint32), outs["hot"].view(np.uint32)), "hot-cache kernel differs from the wave-tile kernel"
    if t == "FLOAT":
        a, b = outs["sell"].astype(np.float64), outs["csr"].astype(np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b)), "sliced-ELL kernel differs from the CSR kernels"
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.all(np.abs(a[fin] - b[fin]) <= 1e-5 * np.maximum(np.abs(b[fin]), 1e-30) + 1e-30), "sliced-ELL kernel differs from the CSR kernels"
    else:
        assert np.array_equal(outs["sell"], outs["csr"]), "sliced-ELL kernel differs from the CSR kernels"
    return outs["sell"]
