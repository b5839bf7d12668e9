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
    """Run the masked SpMV twice — the product path (non-persistent wave-tile kernel) and the experimental persistent
    LDS-hot-cache kernel (SPLAHIP_HOT=1) — assert both give bit-identical results (same work items, same fold order) and
    return the product path's result."""
    from helpers import NP
    dAp, dAj, dAx, dm, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(mask, NP[t])), dev(np.asarray(v, NP[t]))
    import os
    outs = []
    for hot in (False, True):
        plan = kernels.CsrPlan(n_rows, len(Aj), dAp, n_cols if hot else 0, dAj if hot else None)
        r = torch.empty(n_rows, dtype=kernels.torch_dtype(t), device="cuda")
        if junk is not None:
            r.fill_(junk)
        os.environ["SPLAHIP_HOT"] = "1" if hot else "0"  # the persistent kernel is experimental: switched on per call here
        try:
            kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dm, dv, r, op_mult, op_add, op_select, init, early_exit)
            torch.cuda.synchronize()
        finally:
            os.environ["SPLAHIP_HOT"] = "0"
        outs.append(host(r, NP[t]))
        plan.close()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), "hot-cache kernel differs from the wave-tile kernel"
    return outs[0]
