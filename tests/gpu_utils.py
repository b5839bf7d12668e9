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
    from helpers import NP
    dAp, dAj, dAx, dm, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(mask, NP[t])), dev(np.asarray(v, NP[t]))
    plan = kernels.CsrPlan(n_rows, len(Aj), dAp)
    r = torch.empty(n_rows, dtype=kernels.torch_dtype(t), device="cuda")
    if junk is not None:
        r.fill_(junk)
    kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dm, dv, r, op_mult, op_add, op_select, init, early_exit)
    torch.cuda.synchronize()
    out = host(r, NP[t])
    plan.close()
    return out
