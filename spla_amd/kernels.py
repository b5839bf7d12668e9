"""Python face of the kernel-level C ABI for device-resident operands held in torch tensors.

PyTorch-ROCm is plumbing only (device memory, the current HIP stream, torch.distributed); every compute call goes
through libspla_amd.so.  Tensors are passed as raw device pointers (`tensor.data_ptr()`), int32 tensors stand in
for uint32 words where torch has no unsigned dtype.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .ops import binop_id, scalar_bits, selop_id, type_id

_TORCH_DTYPE = {0: torch.int32, 1: torch.int32, 2: torch.float32}  # uint32 words travel as int32 bit patterns


def torch_dtype(t):
    return _TORCH_DTYPE[type_id(t)]


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor expected"
    return C.c_void_p(t.data_ptr())


def current_stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class CsrPlan:
    """Wave-tile work decomposition of one device CSR matrix (splahip_csr_plan_*)."""

    def __init__(self, n_rows: int, nnz: int, Ap: torch.Tensor, n_cols: int = 0, Aj: torch.Tensor = None, Ax: torch.Tensor = None):
        """Ap alone gives the wave-tile decomposition of the CSR kernels.  With (n_cols, Aj, Ax) the plan also builds the
        sliced-ELL + hot-column copy that mxv uses for associative op_add (attach_sell)."""
        self.n_rows, self.nnz = int(n_rows), int(nnz)
        self.sell = None
        self.cob = None
        self._h = C.c_void_p()
        _lib.call("splahip_csr_plan_create", C.byref(self._h), self.n_rows, self.nnz, _ptr(Ap), current_stream_ptr())
        nb, nl = C.c_uint32(), C.c_uint32()
        _lib.call("splahip_csr_plan_info", self._h, C.byref(nb), C.byref(nl))
        self.n_items, self.n_long = nb.value, nl.value
        self.n_blocks = self.n_items  # work items (one per wavefront)
        if Aj is not None and Ax is not None and n_cols:
            self.attach_sell(n_cols, Ap, Aj, Ax)

    def attach_sell(self, n_cols: int, Ap: torch.Tensor, Aj: torch.Tensor, Ax: torch.Tensor):
        """sliced-ELL + hot-column copy of the matrix (values included: rebuild when the matrix changes)"""
        _lib.call("splahip_csr_plan_attach_sell", self._h, int(n_cols), _ptr(Ap), _ptr(Aj) if self.nnz else None, _ptr(Ax) if self.nnz else None,
                  current_stream_ptr())
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.call("splahip_csr_plan_sell_info", self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        u, xb = C.c_int(), C.c_uint32()
        _lib.call("splahip_csr_plan_sell_values", self._h, C.byref(u), C.byref(xb))
        self.sell = {"slices": a.value, "slots": b.value, "hot_columns": c.value, "long_rows": d.value, "uniform_values": bool(u.value)} if a.value else None

    def attach_cob(self, n_cols: int, Ap: torch.Tensor, Aj: torch.Tensor, Ax: torch.Tensor, row_bits: int = 0, tile_chunks: int = 0):
        """column-ordered-blocks copy of the matrix (values included: rebuild when the matrix changes); mxv then uses it for every
        op_add with an LDS atomic.  row_bits / tile_chunks: 0 = the library's defaults."""
        _lib.call("splahip_csr_plan_attach_cob", self._h, int(n_cols), _ptr(Ap), _ptr(Aj) if self.nnz else None, _ptr(Ax) if self.nnz else None,
                  int(row_bits), int(tile_chunks), current_stream_ptr())
        a, b, c, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.call("splahip_csr_plan_cob_info", self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        nr, nh, hr, by = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        _lib.call("splahip_csr_plan_cob_layout", self._h, C.byref(nr), C.byref(nh), C.byref(hr), C.byref(by))
        hs, ns = C.c_uint32(), C.c_uint32()
        _lib.call("splahip_csr_plan_cob_slots", self._h, C.byref(hs), C.byref(ns))
        hot = C.c_uint32()
        _lib.call("splahip_csr_plan_cob_relabel", self._h, C.byref(hot))
        self.cob = {"tiles": a.value, "chunks": b.value, "ranges_in_pieces": c.value, "slots_per_range": d.value, "natural_slots": ns.value, "ranges": nr.value,
                    "hub_ranges": nh.value, "hub_rows": hr.value, "bytes": by.value, "relabelled_hot_columns": hot.value} if a.value else None
        return self.cob

    def mask_probe(self):
        """what the last masked call over the blocks found when it probed its mask on the device (blocking): {"selected_entries", "served_by_rows"}"""
        se, by = C.c_uint64(), C.c_int()
        _lib.call("splahip_csr_plan_cob_probe", self._h, C.byref(se), C.byref(by))
        return {"selected_entries": se.value, "served_by_rows": bool(by.value)}

    def set_uniform_values(self, is_uniform: bool, value_bits: int = 0):
        """the owner of the value array says: every stored value is the word `value_bits` (or not)"""
        _lib.call("splahip_csr_plan_set_uniform_values", self._h, 1 if is_uniform else 0, int(value_bits) & 0xFFFFFFFF)

    def hint_products(self, products: int):
        """products of the next vxm_pull over this (transposed) plan (vxm_probe's count)"""
        _lib.call("splahip_csr_plan_hint_products", self._h, int(products))

    def first_hit(self):
        """what the plan's last BFS-like call did (mxv_first_hit.hpp): {"taken": ran on the first-hit kernel, "uniform": verdict of the last finished pass over
        v (None: none yet), "nonzeros": its non-zero count, "tail_pct": share of the sampled selected entries in rows the walk would enter past their fourth entry}"""
        t, u, n, p = C.c_int(), C.c_int(), C.c_uint32(), C.c_uint32()
        _lib.call("splahip_csr_plan_first_hit", self._h, C.byref(t), C.byref(u), C.byref(n), C.byref(p))
        return {"taken": bool(t.value), "uniform": None if u.value < 0 else bool(u.value), "nonzeros": n.value, "tail_pct": p.value}

    def footprint(self):
        """device bytes held for the matrix besides its CSR arrays: {"plan", "sliced_ell", "column_ordered_blocks"}"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.call("splahip_csr_plan_footprint", self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"plan": a.value, "sliced_ell": b.value, "column_ordered_blocks": c.value}

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            _lib.call("splahip_csr_plan_destroy", self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mxv_masked(plan: CsrPlan, t, n_cols, Ap, Aj, Ax, mask, v, r, op_mult, op_add, op_select, init, early_exit=False):
    """r = mask-selected rows of (M x v) over (op_mult, op_add), written for every row (splahip_mxv_masked)."""
    tid = type_id(t)
    _lib.call("splahip_mxv_masked", plan.handle, tid, plan.n_rows, int(n_cols), _ptr(Ap), _ptr(Aj), _ptr(Ax),
              _ptr(mask), _ptr(v), _ptr(r), binop_id(op_mult), binop_id(op_add), selop_id(op_select),
              scalar_bits(init, tid), 1 if early_exit else 0, current_stream_ptr())
    return r


def mxv_variant(t, op_mult, op_add, early_exit=False, plan: CsrPlan = None) -> str:
    """name of the kernel splahip_mxv_masked picks (with `plan`: its sliced-ELL copy taken into account)"""
    if plan is not None:
        return _lib.call("splahip_mxv_variant_of", plan.handle, type_id(t), binop_id(op_mult), binop_id(op_add), 1 if early_exit else 0).decode()
    return _lib.call("splahip_mxv_variant", type_id(t), binop_id(op_mult), binop_id(op_add), 1 if early_exit else 0).decode()


def csr_transpose(n_rows, n_cols, Ap, Aj, Ax):
    """Device CSR of the transpose (rows ascending inside every column): returns (Tp, Tj, Tx)."""
    nnz = int(Aj.numel())
    Tp = torch.empty(n_cols + 1, dtype=torch.int32, device=Ap.device)
    Tj = torch.empty(max(nnz, 1), dtype=torch.int32, device=Ap.device)[:nnz]
    Tx = torch.empty(max(nnz, 1), dtype=Ax.dtype, device=Ap.device)[:nnz]
    _lib.call("splahip_csr_transpose", int(n_rows), int(n_cols), nnz, _ptr(Ap), _ptr(Aj) if nnz else None, _ptr(Ax) if nnz else None,
              _ptr(Tp), _ptr(Tj) if nnz else None, _ptr(Tx) if nnz else None, current_stream_ptr())
    return Tp, Tj, Tx


def vxm_probe(vi, Ap):
    prod, srt = C.c_uint64(), C.c_int()
    _lib.call("splahip_vxm_probe", int(vi.numel()), _ptr(vi) if vi.numel() else None, _ptr(Ap), C.byref(prod), C.byref(srt), current_stream_ptr())
    return prod.value, bool(srt.value)


def vxm_pull(plan_T: CsrPlan, t, n_rows, n_cols, Tp, Tj, Tx, mask, vi, vx, op_mult, op_add, op_select, identity=False):
    """Pull form of vxm over the transposed CSR; returns (ri, rx) index-sorted COO tensors."""
    tid = type_id(t)
    ri = torch.empty(max(n_cols, 1), dtype=torch.int32, device=mask.device)
    rx = torch.empty(max(n_cols, 1), dtype=torch_dtype(t), device=mask.device)
    cnt = C.c_uint32()
    _lib.call("splahip_vxm_pull", plan_T.handle, tid, int(n_rows), int(n_cols), _ptr(Tp), _ptr(Tj), _ptr(Tx), _ptr(mask),
              int(vi.numel()), _ptr(vi), _ptr(vx), 1 if identity else 0, _ptr(ri), _ptr(rx), C.byref(cnt),
              binop_id(op_mult), binop_id(op_add), selop_id(op_select), current_stream_ptr())
    return ri[:cnt.value], rx[:cnt.value]


def vxm_push(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, op_mult, op_add, op_select):
    """Push form of vxm (expand / stable sort / fold); returns (ri, rx) index-sorted COO tensors."""
    tid = type_id(t)
    ri = torch.empty(max(n_cols, 1), dtype=torch.int32, device=mask.device)
    rx = torch.empty(max(n_cols, 1), dtype=torch_dtype(t), device=mask.device)
    cnt = C.c_uint32()
    _lib.call("splahip_vxm_push", tid, int(n_rows), int(n_cols), _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(mask),
              int(vi.numel()), _ptr(vi) if vi.numel() else None, _ptr(vx) if vi.numel() else None, _ptr(ri), _ptr(rx), C.byref(cnt),
              binop_id(op_mult), binop_id(op_add), selop_id(op_select), current_stream_ptr())
    return ri[:cnt.value], rx[:cnt.value]


def vxm_push_small(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, op_mult, op_add, op_select, max_products):
    """Single-synchronisation push for small frontiers: returns (done, products, sorted, ri, rx); when the device-side product count
    exceeds max_products nothing is computed (done = False) and the caller falls back to vxm_push / vxm_pull."""
    tid = type_id(t)
    ri = torch.empty(max(n_cols, 1), dtype=torch.int32, device=mask.device)
    rx = torch.empty(max(n_cols, 1), dtype=torch_dtype(t), device=mask.device)
    cnt, products, srt, done = C.c_uint32(), C.c_uint64(), C.c_int(), C.c_int()
    _lib.call("splahip_vxm_push_small", tid, int(n_rows), int(n_cols), _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(mask),
              int(vi.numel()), _ptr(vi) if vi.numel() else None, _ptr(vx) if vi.numel() else None, _ptr(ri), _ptr(rx), int(max_products), C.byref(cnt),
              C.byref(products), C.byref(srt), C.byref(done), binop_id(op_mult), binop_id(op_add), selop_id(op_select), current_stream_ptr())
    return bool(done.value), int(products.value), bool(srt.value), ri[:cnt.value], rx[:cnt.value]


def _adopt(ptr, n, dtype):
    """Copy a library-allocated device array of n words into a torch tensor and release the original."""
    out = torch.empty(max(n, 1), dtype=dtype, device="cuda")[:n]
    if n:
        _lib.call("splahip_memcpy_d2d", _ptr(out), ptr, n * 4, current_stream_ptr())
        torch.cuda.current_stream().synchronize()
    if ptr:
        _lib.call("splahip_free", ptr)
    return out


def mxmT_masked(t, n_rows, M, A, B, op_mult, op_add, op_select, init):
    """R = (A x B^T) .mask; M, A, B are (Ap, Aj, Ax) device CSR triples; returns the CSR triple of R."""
    tid = type_id(t)
    Rp = torch.empty(n_rows + 1, dtype=torch.int32, device="cuda")
    pj, px, nnz = C.c_void_p(), C.c_void_p(), C.c_uint32()
    mask_nnz = int(M[1].numel())
    args = []
    for (p_, j_, x_) in (M, A, B):
        args += [_ptr(p_), _ptr(j_) if j_.numel() else None, _ptr(x_) if x_.numel() else None]
    _lib.call("splahip_mxmT_masked", tid, int(n_rows), mask_nnz, *args, _ptr(Rp), C.byref(pj), C.byref(px), C.byref(nnz),
              binop_id(op_mult), binop_id(op_add), selop_id(op_select), scalar_bits(init, tid), current_stream_ptr())
    return Rp, _adopt(pj, nnz.value, torch.int32), _adopt(px, nnz.value, torch_dtype(t))


class LibraryCsr:
    """A CSR result whose column / value arrays are still the library's allocations (no copy): `.nnz`, `.Rp` (tensor), raw
    device pointers `.pj / .px`; `.adopt(t)` copies them into torch tensors, `.free()` releases them (splahip_free)."""

    def __init__(self, Rp, pj, px, nnz):
        self.Rp, self.pj, self.px, self.nnz = Rp, pj, px, int(nnz)

    def adopt(self, t):
        out = (self.Rp, _adopt(self.pj, self.nnz, torch.int32), _adopt(self.px, self.nnz, torch_dtype(t)))
        self.pj = self.px = None
        return out

    def free(self):
        for q in (self.pj, self.px):
            if q:
                _lib.call("splahip_free", q)
        self.pj = self.px = None


def mxm(t, a_rows, b_cols, A, B, op_mult, op_add, init, block_products=0, adopt=True):
    """R = A x B (ESC); returns ((Rp, Rj, Rx), products), or (LibraryCsr, products) with adopt=False (what a C host sees:
    the call itself, without this wrapper's copy of the result into torch tensors)."""
    tid = type_id(t)
    Rp = torch.empty(a_rows + 1, dtype=torch.int32, device="cuda")
    pj, px, nnz, prod = C.c_void_p(), C.c_void_p(), C.c_uint32(), C.c_uint64()
    args = []
    for (p_, j_, x_) in (A, B):
        args += [_ptr(p_), _ptr(j_) if j_.numel() else None, _ptr(x_) if x_.numel() else None]
    _lib.call("splahip_mxm", tid, int(a_rows), int(b_cols), *args, _ptr(Rp), C.byref(pj), C.byref(px), C.byref(nnz), C.byref(prod),
              binop_id(op_mult), binop_id(op_add), scalar_bits(init, tid), int(block_products), current_stream_ptr())
    raw = LibraryCsr(Rp, pj, px, nnz.value)
    return (raw.adopt(t) if adopt else raw), prod.value
