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


_ORACLE = None


def _oracle():
    global _ORACLE
    if _ORACLE is None:
        from oracle import pyoracle
        _ORACLE = pyoracle.Oracle()
    return _ORACLE


def _same_bits_or_both_nan(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype != np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def _assert_form_matches_oracle(form, out, want, t, op_add, early_exit):
    """One kernel form against the ORACLE directly (not against another form).  int / uint, every selecting fold and every fold the form keeps in the reference's
    order: the reference's word.  An fp32 PLUS fold that a form re-associates: north_star's 1e-5 relative, with the same NaN / inf pattern."""
    if t != "FLOAT":
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), f"{form}: differs from the oracle"
        return
    if op_add != "PLUS" or early_exit or form == "csr-ref":
        assert _same_bits_or_both_nan(out, want), f"{form}: fp32 {op_add} differs bit-wise from the oracle"
        return
    a, b = out.astype(np.float64), want.astype(np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b)), f"{form}: NaN / inf pattern differs from the oracle"
    fin = np.isfinite(a) & np.isfinite(b)
    assert np.all(np.abs(a[fin] - b[fin]) <= 1e-5 * np.abs(b[fin]) + 1e-30), f"{form}: fp32 sum beyond 1e-5 relative of the oracle"


def hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, op_mult, op_add, op_select, init, early_exit=False, junk=None, exact=False, calls=1, fp32_sum_vs_oracle=True):
    """Run the masked SpMV on EVERY kernel form — the CSR wave-tile / row kernels (what the operator API uses), the same call in reference-order mode
    (splahip_set_reference_order), the opt-in sliced-ELL copy (value form and forced bit form), the column-ordered blocks (default geometry, a small geometry
    that exercises shared accumulators, and both on the run-time schedule) — and compare each form with the ORACLE directly: bit-identical for int / uint, for
    every selecting fold and for the reference-order mode; within 1e-5 relative for an fp32 PLUS fold a form re-associates.  Returns the CSR result.
    fp32_sum_vs_oracle=False (inputs whose fp32 sums cancel: the relative bar is then not the contract): the re-associating forms are compared with the CSR
    form as before, the reference-order form still bit for bit with the oracle."""
    from helpers import NP
    from spla_amd import _lib
    dAp, dAj, dAx, dm, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(mask, NP[t])), dev(np.asarray(v, NP[t]))
    import os
    want = _oracle().mxv_masked(t, n_rows, n_cols, np.ascontiguousarray(Ap), np.ascontiguousarray(Aj), np.asarray(Ax, NP[t]), np.asarray(mask, NP[t]),
                                np.asarray(v, NP[t]), op_mult, op_add, op_select, init, early_exit)
    outs = {}
    # sell-bits: the sparse-input-vector form forced on wherever the semiring allows it (SPLAHIP_SELL_BITS=2); plain "sell" lets the
    # device statistics choose
    for form in ("csr", "csr-ref", "sell", "sell-bits", "cob", "cob-small", "cob-dyn", "cob-small-dyn"):
        os.environ["SPLAHIP_SELL_BITS"] = "2" if form == "sell-bits" else "1"
        # cob-small also forces cob's sparse-input-vector form of min-plus, which the library only picks for >= 2^23 columns
        os.environ["SPLAHIP_COB_BITS"] = "2" if form in ("cob-small", "cob-small-dyn") else "1"
        plan = kernels.CsrPlan(n_rows, len(Aj), dAp, n_cols, dAj, dAx) if form.startswith("sell") else kernels.CsrPlan(n_rows, len(Aj), dAp)
        # "-dyn": the run-time schedule of the blocks kernel (mxv_cob_dyn.hpp: persistent workgroups that join ranges and claim rounds; shipped switched
        # off) — on the small geometry with three listed workers per range so that ranges are shared, merged and finished by the kernel behind the launch
        if form.endswith("-dyn"):
            os.environ["SPLAHIP_COB_DYN"] = "1"
            if form == "cob-small-dyn":
                os.environ["SPLAHIP_COB_DYN_WORKERS"] = "3"
        if form in ("cob", "cob-dyn"):  # column-ordered blocks, default geometry (ranges of 20224 slots)
            plan.attach_cob(n_cols, dAp, dAj, dAx)
        elif form in ("cob-small", "cob-small-dyn"):  # ranges of 256 slots, pieces of 2 chunks, every row of degree >= 3 a hub candidate: hub and natural ranges, column
            os.environ["SPLAHIP_COB_HUB_MIN_DEGREE"] = "3"  # windows, ranges in several pieces (shared accumulators, tickets) even on small inputs
            os.environ["SPLAHIP_COB_NAT_SLOTS"] = "512"     # natural ranges twice the hub ranges' size: an fp32 sum keeps them in float words (COB_FADD)
            plan.attach_cob(n_cols, dAp, dAj, dAx, row_bits=8, tile_chunks=2)
            os.environ.pop("SPLAHIP_COB_HUB_MIN_DEGREE")
            os.environ.pop("SPLAHIP_COB_NAT_SLOTS")
        os.environ.pop("SPLAHIP_COB_DYN", None)
        os.environ.pop("SPLAHIP_COB_DYN_WORKERS", None)
        r = torch.empty(n_rows, dtype=kernels.torch_dtype(t), device="cuda")
        if form == "csr-ref":
            assert _lib.call("splahip_set_reference_order", 1) == 0
        try:
            for _ in range(calls):  # calls > 1: the later calls of a plan follow the hints its earlier calls left (mask probe, first-hit pass); the last result is checked
                if junk is not None:
                    r.fill_(junk)
                kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dm, dv, r, op_mult, op_add, op_select, init, early_exit)
                torch.cuda.synchronize()
        finally:
            if form == "csr-ref":
                assert _lib.call("splahip_set_reference_order", 0) == 1
        outs[form] = host(r, NP[t])
        plan.close()
    os.environ["SPLAHIP_SELL_BITS"] = "1"
    os.environ["SPLAHIP_COB_BITS"] = "1"
    # every form against the oracle itself
    for form, out in outs.items():
        if t == "FLOAT" and op_add == "PLUS" and not early_exit and not fp32_sum_vs_oracle and form != "csr-ref":
            continue
        _assert_form_matches_oracle(form, out, want, t, op_add, early_exit)
    if exact:  # folds that select (MIN / MAX): every form must give the reference's word, NaN payloads and all
        for form, out in outs.items():
            assert np.array_equal(out.view(np.uint32), outs["csr"].view(np.uint32)), f"{form} differs bit-wise from the CSR kernels"
    assert np.array_equal(outs["sell-bits"].view(np.uint32), outs["sell"].view(np.uint32)), "sliced-ELL sparse-vector form differs from the value form"
    if t == "FLOAT":
        a, b = outs["sell"].astype(np.float64), outs["csr"].astype(np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b)), "sliced-ELL kernel differs from the CSR kernels"
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.all(np.abs(a[fin] - b[fin]) <= 1e-5 * np.maximum(np.abs(b[fin]), 1e-30) + 1e-30), "sliced-ELL kernel differs from the CSR kernels"
    else:
        assert np.array_equal(outs["sell"], outs["csr"]), "sliced-ELL kernel differs from the CSR kernels"
    for form in ("cob", "cob-small", "cob-dyn", "cob-small-dyn"):
        if t == "FLOAT" and op_add in ("MIN", "MAX") and op_mult == "PLUS":  # the keyed fp32 MIN / MAX form is the reference's fold bit for bit
            assert np.array_equal(outs[form].view(np.uint32), outs["csr"].view(np.uint32)), f"{form} fp32 {op_add} differs from the CSR kernels"
        elif t == "FLOAT":
            a, b = outs[form].astype(np.float64), outs["csr"].astype(np.float64)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b)), f"{form} kernel differs from the CSR kernels"
            fin = np.isfinite(a) & np.isfinite(b)
            assert np.all(np.abs(a[fin] - b[fin]) <= 1e-5 * np.maximum(np.abs(b[fin]), 1e-30) + 1e-30), f"{form} kernel differs from the CSR kernels"
        else:
            assert np.array_equal(outs[form], outs["csr"]), f"{form} (column-ordered blocks) kernel differs from the CSR kernels"
    return outs["csr"]
