"""1-D row-block decomposition of the masked SpMV across the ranks of one node.

* rows are cut into `world` contiguous blocks that balance nnz (prefix sum over Ap), not row count;
* rank p owns the CSR slice of its rows (column ids stay global), and slice p of every vector under the same cuts;
* one exchange step per SpMV: the all-gather of the dense input vector.  RCCL has no all-gatherv, so slices are padded
  to the longest block and exchanged with ONE all_gather_into_tensor; the local CSR's column ids are remapped once,
  at setup, to index that padded [world][max_rows] layout directly, so the step does no unpacking.
* sparsity-aware form of the same step (`SparseExchange`) for vectors that are mostly fill (a BFS frontier, the changed
  distances of an SSSP round): every rank sends only the positions (and values) of its non-fill entries, padded to the
  longest list; the receivers clear the previous round's positions and scatter the new ones.  At 8 ranks and a 2 %
  frontier that is ~1 MB instead of 134 MB per step.

Everything here is device-agnostic torch + torch.distributed (nccl == RCCL on ROCm; gloo in the CPU tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def row_splits(Ap: torch.Tensor, world: int):
    """world + 1 row indices that balance nnz across the blocks."""
    from .graphgen import nnz_balanced_row_splits
    return nnz_balanced_row_splits(Ap, world)


class RowBlock:
    """This rank's block of an n x n matrix given as global CSR (Ap, Aj[, Ax]) tensors."""

    def __init__(self, Ap, Aj, Ax, bounds, rank):
        self.bounds, self.rank, self.world = list(bounds), rank, len(bounds) - 1
        self.n = int(Ap.numel() - 1)
        self.lo, self.hi = self.bounds[rank], self.bounds[rank + 1]
        self.n_local = self.hi - self.lo
        s, e = int(Ap[self.lo]), int(Ap[self.hi])
        self.Ap = (Ap[self.lo:self.hi + 1] - Ap[self.lo]).contiguous()
        self.Aj_global = Aj[s:e].clone()
        self.Ax = Ax[s:e].clone() if Ax is not None else None
        self.nnz_local = e - s
        self.counts = [self.bounds[p + 1] - self.bounds[p] for p in range(self.world)]
        self.max_rows = max(self.counts)
        self.n_cols_padded = self.world * self.max_rows if self.world > 1 else self.n
        self.Aj = self.remap_columns(self.Aj_global) if self.world > 1 else self.Aj_global

    def remap_columns(self, cols: torch.Tensor) -> torch.Tensor:
        """global column id -> index into the padded all-gathered vector (owner * max_rows + offset in the owner's slice)."""
        b = torch.tensor(self.bounds, device=cols.device, dtype=torch.int64)
        c = cols.to(torch.int64)
        owner = torch.searchsorted(b[1:].contiguous(), c, right=True)
        return (owner * self.max_rows + (c - b[owner])).to(cols.dtype)

    def pad_local(self, x_local: torch.Tensor, fill=0) -> torch.Tensor:
        """this rank's slice, padded to max_rows (what the all-gather sends)."""
        if self.world == 1:
            return x_local
        out = torch.full((self.max_rows,), fill, dtype=x_local.dtype, device=x_local.device)
        out[:self.n_local] = x_local
        return out

    def all_gather(self, x_local_padded: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """the exchange step: every rank contributes its padded slice, every rank receives [world][max_rows]."""
        if self.world == 1:
            return x_local_padded
        if out is None:
            out = torch.empty(self.n_cols_padded, dtype=x_local_padded.dtype, device=x_local_padded.device)
        if x_local_padded.is_cuda and dist.get_backend() == "gloo":
            # debug path (two ranks sharing one GPU, SPLA_DIST_BACKEND=gloo): stage the exchange through the host
            parts = [torch.empty(self.max_rows, dtype=out.dtype) for _ in range(self.world)]
            dist.all_gather(parts, x_local_padded.cpu())
            out.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(out, x_local_padded)
        return out

    def unpad(self, x_padded: torch.Tensor) -> torch.Tensor:
        """padded [world][max_rows] layout -> the global vector (test / host read-back helper)."""
        if self.world == 1:
            return x_padded
        parts = [x_padded[p * self.max_rows: p * self.max_rows + self.counts[p]] for p in range(self.world)]
        return torch.cat(parts)


class SparseExchange:
    """Sparsity-aware exchange of this rank's slice into the padded gathered vector `v_full` (len n_cols_padded + 1: the last word is
    a scratch slot that padding positions point to).  One call = one full exchange of a fresh vector: local compaction, a count
    exchange (lists are padded to the longest), one all-gather of positions (+ one of values unless `value` is given), clearing of
    the previous round's positions, scatter.  Result: v_full[:n_cols_padded] equals what RowBlock.all_gather would produce."""

    def __init__(self, blk: RowBlock, dtype, device, fill=0):
        self.blk, self.fill = blk, fill
        self.trash = blk.n_cols_padded
        assert blk.n_cols_padded < 2 ** 31, "positions are exchanged as 4-byte words"
        self.v_full = torch.full((blk.n_cols_padded + 1,), fill, dtype=dtype, device=device)
        self.prev = None  # positions written by the previous exchange
        self.last_count = 0  # entries exchanged by the last call (all ranks)

    def _all_gather(self, out: torch.Tensor, x: torch.Tensor):
        if x.is_cuda and dist.get_backend() == "gloo":  # debug path: ranks sharing one GPU, exchange staged through the host
            parts = [torch.empty(x.numel(), dtype=x.dtype) for _ in range(self.blk.world)]
            dist.all_gather(parts, x.cpu())
            out.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(out, x)

    def exchange(self, x_local: torch.Tensor, value=None, upper_bound: int = None) -> torch.Tensor:
        """x_local: this rank's n_local entries.  value: if every non-fill entry holds this value (a 0/1 frontier), values are not sent.
        upper_bound: a number every rank knows that bounds each rank's non-fill count (e.g. the global frontier size the previous
        iteration counted): then lists are padded to it and neither the count exchange nor any host synchronisation is needed."""
        blk, dev = self.blk, x_local.device
        live = x_local != self.fill
        if upper_bound is not None and hasattr(torch, "nonzero_static"):
            m = max(int(min(upper_bound, blk.max_rows)), 1)
            if os.environ.get("SPLA_CHECK_BOUNDS") == "1":  # debug: nonzero_static truncates silently when the caller's bound is too small
                assert int(live.sum()) <= m, "SparseExchange: upper_bound is smaller than this rank's non-fill count"
            idx = torch.nonzero_static(live, size=m, fill_value=-1).squeeze(1)  # ascending local row ids, -1 beyond the count: no sync
            send = torch.where(idx >= 0, idx + blk.rank * blk.max_rows, torch.full_like(idx, self.trash))
            total = None
        else:
            idx = torch.nonzero(live).squeeze(1)  # (host sync: the list length)
            c = int(idx.numel())
            cnt = torch.tensor([c], dtype=torch.int64, device=dev)
            cnts = torch.empty(blk.world, dtype=torch.int64, device=dev)
            self._all_gather(cnts, cnt)
            counts = cnts.tolist()
            m = max(max(counts), 1)
            send = torch.full((m,), self.trash, dtype=torch.int64, device=dev)
            send[:c] = idx + blk.rank * blk.max_rows  # position in the padded [world][max_rows] layout
            idx = torch.cat([idx, torch.full((m - c,), -1, dtype=torch.int64, device=dev)])
            total = sum(counts)
        # positions travel as 4-byte words (n_cols_padded < 2^31), half the bytes of the index type torch scatters with
        recv32 = torch.empty(blk.world * m, dtype=torch.int32, device=dev)
        self._all_gather(recv32, send.to(torch.int32))
        recv = recv32.to(torch.int64)
        if self.prev is not None:
            self.v_full.index_fill_(0, self.prev, self.fill)
        if value is not None:
            self.v_full.index_fill_(0, recv, value)
        else:
            sv = torch.where(idx >= 0, x_local[idx.clamp_min(0)], torch.full((m,), self.fill, dtype=x_local.dtype, device=dev))
            rv = torch.empty(blk.world * m, dtype=x_local.dtype, device=dev)
            self._all_gather(rv, sv)
            self.v_full.index_put_((recv,), rv)
        self.v_full[self.trash] = self.fill
        self.prev = recv
        self.last_count = total  # None when the caller supplied the bound (the count is then the caller's knowledge)
        return self.v_full


class SparseExchangeABI:
    """The sparsity-aware exchange behind the kernel-level C ABI (splahip_dist_sx_* / splahip_dist_exchange_sparse, dist.hip): a rank's message is
    [count | m positions in the padded layout | m values], m = a bound every rank knows for this exchange; the receivers return the previous exchange's
    positions to the fill value and write the new ones.  Same contract as SparseExchange (v_full[:n_cols_padded] equals the dense all-gather).
      * device tensors + a DistComm: one C-ABI call (pack, RCCL all-gather, scatter on the caller's stream);
      * device tensors without a communicator (gloo debug run, ranks sharing a GPU): the C ABI's pack / scatter kernels around a host-staged
        torch.distributed all-gather of the messages;
      * CPU tensors (the gloo tests of this container, which has no GPU): `_pack_host` / `_scatter_host` below mirror the two kernels word for word
        on numpy-style torch ops, so the message layout, the bound and the clearing of the previous round are what the tests exercise."""

    def __init__(self, blk: RowBlock, dtype, device, fill=0, capacity: int = None, comm: "DistComm" = None):
        self.blk, self.fill, self.comm = blk, fill, comm
        self.max_rows = blk.max_rows if blk.world > 1 else blk.n
        self.capacity = int(min(capacity or self.max_rows, self.max_rows))
        self.v_full = torch.full((blk.n_cols_padded,), fill, dtype=dtype, device=device)
        self.on_device = torch.device(device).type == "cuda"
        self.last_words = 0
        self._prev = None  # host mirror: (recv messages, m, with_values) of the previous exchange
        self._h = None
        if self.on_device:
            import ctypes as C
            from . import _lib
            self._h = C.c_void_p()
            _lib.call("splahip_dist_sx_create", C.byref(self._h), blk.world, blk.rank, self.max_rows, self.capacity)

    @staticmethod
    def _bits(x, dtype):
        return int(torch.tensor([x], dtype=dtype).view(torch.int32)[0]) & 0xFFFFFFFF

    # ---- host mirror of sx_pack_kernel / sx_scatter_kernel (CPU tensors) ----
    def _pack_host(self, x_local, m, with_values):
        words = 1 + (2 if with_values else 1) * m
        msg = torch.zeros(words, dtype=torch.int32)
        idx = torch.nonzero(x_local != self.fill).squeeze(1)
        msg[0] = int(idx.numel())
        k = min(int(idx.numel()), m)  # entries beyond the bound are dropped and reported (overflow)
        self.overflow = self.overflow or int(idx.numel()) > m
        base = self.blk.rank * self.max_rows if self.blk.world > 1 else 0
        msg[1:1 + k] = (idx[:k] + base).to(torch.int32)
        if with_values:
            msg[1 + m:1 + m + k] = x_local[idx[:k]].view(torch.int32) if x_local.dtype != torch.int32 else x_local[idx[:k]]
        return msg

    def _scatter_host(self, recv, m, layout_values, write_values, value_i32):
        """rank p's message = recv[p * words:(p + 1) * words]; its first min(count, m) positions are written: with the message's values (write_values)
        or with the one word value_i32 (a value-less exchange; the clearing of the previous round)"""
        words = 1 + (2 if layout_values else 1) * m
        vf = self.v_full.view(torch.int32) if self.v_full.dtype != torch.int32 else self.v_full
        for p in range(self.blk.world):
            msg = recv[p * words:(p + 1) * words]
            k = min(int(msg[0]), m)
            pos = msg[1:1 + k].to(torch.int64)
            vf[pos] = msg[1 + m:1 + m + k] if write_values else torch.full((k,), value_i32, dtype=torch.int32)

    def exchange(self, x_local: torch.Tensor, value=None, upper_bound: int = None, direct=False) -> torch.Tensor:
        blk = self.blk
        m = int(min(upper_bound or self.capacity, self.capacity))
        with_values = value is None
        words = 1 + (2 if with_values else 1) * m
        self.last_words = words
        fill_bits = self._bits(self.fill, self.v_full.dtype)
        value_bits = 0 if with_values else self._bits(value, self.v_full.dtype)
        if not self.on_device:
            self.overflow = getattr(self, "overflow", False)
            send = self._pack_host(x_local, m, with_values)
            recv = torch.empty(blk.world * words, dtype=torch.int32)
            if blk.world > 1:
                dist.all_gather_into_tensor(recv, send)
            else:
                recv.copy_(send)
            i32 = lambda b: b if b < 2 ** 31 else b - 2 ** 32
            if self._prev is not None:
                pr, pm, pw = self._prev
                self._scatter_host(pr, pm, pw, False, i32(fill_bits))
            self._scatter_host(recv, m, with_values, with_values, i32(value_bits))
            self._prev = (recv, m, with_values)
            return self.v_full
        import ctypes as C
        from . import _lib
        from .kernels import _ptr, current_stream_ptr
        if self.comm is not None and blk.world > 1:
            _lib.call("splahip_dist_exchange_sparse", self.comm._h, self._h, _ptr(x_local), int(x_local.numel()), fill_bits, m, 1 if with_values else 0, value_bits,
                      _ptr(self.v_full), 1 if direct else 0, current_stream_ptr())
            return self.v_full
        _lib.call("splahip_dist_sx_pack", self._h, _ptr(x_local), int(x_local.numel()), fill_bits, m, 1 if with_values else 0, current_stream_ptr())
        ps, pr, w = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _lib.call("splahip_dist_sx_buffers", self._h, m, 1 if with_values else 0, C.byref(ps), C.byref(pr), C.byref(w))
        assert w.value == words
        if blk.world > 1:  # host-staged transport between the C-ABI phases (ranks sharing one GPU under the gloo debug backend)
            torch.cuda.synchronize()
            send = torch.empty(words, dtype=torch.int32)
            _lib.call("splahip_memcpy_d2h", send.data_ptr(), ps, 4 * words, None)
            recv = torch.empty(blk.world * words, dtype=torch.int32)
            dist.all_gather_into_tensor(recv, send)
            _lib.call("splahip_memcpy_h2d", pr, recv.data_ptr(), 4 * words * blk.world, None)
        else:
            _lib.call("splahip_memcpy_d2d", pr, ps, 4 * words, current_stream_ptr())
        _lib.call("splahip_dist_sx_scatter", self._h, _ptr(self.v_full), fill_bits, m, 1 if with_values else 0, value_bits, current_stream_ptr())
        return self.v_full

    def overflowed(self) -> bool:
        if not self.on_device:
            o, self.overflow = getattr(self, "overflow", False), False
            return o
        import ctypes as C
        from . import _lib
        f = C.c_int()
        _lib.call("splahip_dist_sx_overflowed", self._h, C.byref(f))
        return bool(f.value)

    def close(self):
        if self._h:
            from . import _lib
            _lib.call("splahip_dist_sx_destroy", self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):  # the device buffers behind the handle go with the object (interpreter shutdown: the library may be gone already)
        try:
            self.close()
        except Exception:
            pass


# ---- the product behind the kernel-level C ABI (spla_amd/csrc/kernels/dist.hip) ------------------------------------------------------------
def split_columns(blk: RowBlock):
    """Torch mirror of splahip_shard_create's column split (CPU tests): the row block cut by column OWNER into the local part
    (columns of this rank's own slice, rebased to [0, max_rows)) and the remote part (padded-layout column ids); a stable partition,
    so both parts keep the rows' entry order.  Returns ((LAp, LAj, LAx), (RAp, RAj, RAx))."""
    lo = blk.rank * blk.max_rows if blk.world > 1 else 0
    hi = lo + (blk.max_rows if blk.world > 1 else blk.n)
    cols = blk.Aj.to(torch.int64)
    loc = (cols >= lo) & (cols < hi)
    rows = torch.repeat_interleave(torch.arange(blk.n_local, device=cols.device), (blk.Ap[1:] - blk.Ap[:-1]).to(torch.int64))
    parts = []
    for sel, rebase in ((loc, lo), (~loc, 0)):
        cnt = torch.bincount(rows[sel], minlength=blk.n_local)
        p = torch.zeros(blk.n_local + 1, dtype=torch.int64, device=cols.device)
        torch.cumsum(cnt, 0, out=p[1:])
        parts.append((p.to(blk.Ap.dtype), (cols[sel] - rebase).to(blk.Aj.dtype), blk.Ax[sel] if blk.Ax is not None else None))
    return parts[0], parts[1]


class DistComm:
    """RCCL communicator of the C ABI (splahip_dist_comm_*); rank 0's unique id travels through torch.distributed (any backend)."""

    def __init__(self, world: int, rank: int):
        import ctypes as C
        from . import _lib
        self.world, self.rank = world, rank
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            _lib.call("splahip_dist_get_unique_id", C.byref(buf))
        if world > 1:
            t = torch.tensor(list(buf), dtype=torch.uint8)
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = t.to(dev)
            dist.broadcast(t, src=0)
            buf = (C.c_ubyte * 128)(*t.cpu().tolist())
        self._h = C.c_void_p()
        _lib.call("splahip_dist_comm_create", C.byref(self._h), world, rank, C.byref(buf))

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor, direct=False):
        from . import _lib
        from .kernels import _ptr, current_stream_ptr
        _lib.call("splahip_dist_all_gather", self._h, _ptr(send), _ptr(recv), int(send.numel()), 1 if direct else 0, current_stream_ptr())
        return recv

    def close(self):
        from . import _lib
        if self._h:
            _lib.call("splahip_dist_comm_destroy", self._h)
            self._h = None


class Shard:
    """This rank's row block behind the C ABI: the column split, both plans and their column-ordered-blocks copies (splahip_shard_*).
    Works without a communicator: the phases (local part / remote part / whole block) are separate calls, so a host that brings its own
    exchange — torch.distributed over gloo in the one-GPU debug run, the sequential two-rank simulation of the GPU tests — drives them."""

    def __init__(self, blk: RowBlock, Ax: torch.Tensor):
        import ctypes as C
        from . import _lib
        from .kernels import _ptr, current_stream_ptr
        self.blk = blk
        self._keep = (blk.Ap, blk.Aj, Ax)  # the shard references the caller's arrays
        self._h = C.c_void_p()
        max_rows = blk.max_rows if blk.world > 1 else blk.n
        _lib.call("splahip_shard_create", C.byref(self._h), blk.world, blk.rank, blk.n_local, max_rows, blk.nnz_local, _ptr(blk.Ap),
                  _ptr(blk.Aj) if blk.nnz_local else None, _ptr(Ax) if blk.nnz_local else None, current_stream_ptr())
        a, b, ok = C.c_uint32(), C.c_uint32(), C.c_int()
        _lib.call("splahip_shard_info", self._h, C.byref(a), C.byref(b), C.byref(ok))
        self.nnz_local_cols, self.nnz_remote_cols, self.split_ok = a.value, b.value, bool(ok.value)

    def serves(self, t, op_add, early_exit=False) -> bool:
        from . import _lib
        from .ops import binop_id, type_id
        return bool(_lib.call("splahip_shard_split_serves", self._h, type_id(t), binop_id(op_add), 1 if early_exit else 0))

    def prefers_whole(self, t, op_add) -> bool:
        """the C-ABI product runs the unsplit block (exchange first) for this call although the two-part form could serve it (two ranks)"""
        from . import _lib
        from .ops import binop_id, type_id
        return bool(_lib.call("splahip_shard_prefers_whole", self._h, type_id(t), binop_id(op_add)))

    def mxv_local(self, t, mask, v_local, r, op_mult, op_add, op_select, init):
        from . import _lib
        from .kernels import _ptr, current_stream_ptr
        from .ops import binop_id, scalar_bits, selop_id, type_id
        tid = type_id(t)
        _lib.call("splahip_shard_mxv_local", self._h, tid, _ptr(mask), _ptr(v_local), _ptr(r), binop_id(op_mult), binop_id(op_add), selop_id(op_select),
                  scalar_bits(init, tid), current_stream_ptr())

    def mxv_remote(self, t, mask, v_full, r, op_mult, op_add, op_select):
        from . import _lib
        from .kernels import _ptr, current_stream_ptr
        from .ops import binop_id, selop_id, type_id
        _lib.call("splahip_shard_mxv_remote", self._h, type_id(t), _ptr(mask), _ptr(v_full), _ptr(r), binop_id(op_mult), binop_id(op_add), selop_id(op_select),
                  current_stream_ptr())

    def mxv_whole(self, t, mask, v_full, r, op_mult, op_add, op_select, init, early_exit=False):
        from . import _lib
        from .kernels import _ptr, current_stream_ptr
        from .ops import binop_id, scalar_bits, selop_id, type_id
        tid = type_id(t)
        _lib.call("splahip_shard_mxv_whole", self._h, tid, _ptr(mask), _ptr(v_full), _ptr(r), binop_id(op_mult), binop_id(op_add), selop_id(op_select),
                  scalar_bits(init, tid), 1 if early_exit else 0, current_stream_ptr())

    def product(self, comm: DistComm, t, mask, v_local, v_full, r, op_mult, op_add, op_select, init, early_exit=False, direct=False) -> bool:
        """one sharded masked SpMV over RCCL (splahip_dist_mxv_masked): returns True when the overlapped two-part form ran"""
        import ctypes as C
        from . import _lib
        from .kernels import _ptr, current_stream_ptr
        from .ops import binop_id, scalar_bits, selop_id, type_id
        tid = type_id(t)
        ov = C.c_int()
        _lib.call("splahip_dist_mxv_masked", comm._h, self._h, tid, _ptr(mask), _ptr(v_local), _ptr(v_full), _ptr(r), binop_id(op_mult), binop_id(op_add),
                  selop_id(op_select), scalar_bits(init, tid), 1 if early_exit else 0, 1 if direct else 0, C.byref(ov), current_stream_ptr())
        return bool(ov.value)

    def close(self):
        from . import _lib
        if self._h:
            _lib.call("splahip_shard_destroy", self._h)
            self._h = None


# ---- SpGEMM across the ranks (mxm by rows of A, mxmT by rows of the mask; B replicated) -------------------------------------------
# R = A x B and R = mask .* (A x B^T) compute every result row from ONE row of A (and of the mask) and the whole of B (reference
# src/cpu/cpu_mxm.hpp:85-114, cpu_mxmT_masked.hpp:87-125): rank p takes a contiguous block of rows, keeps B whole, and its result rows are the same
# words the unsharded call produces — no exchange in the data path.  The blocks balance the WORK (products of mxm, merge steps of mxmT), not the row
# count.  What the callers do with R decides the only collective: tc() needs one all-reduced scalar; a host that wants the whole R gathers the row
# blocks (`gather_csr_rows`, for validation and the tests).
def work_splits(cost: torch.Tensor, parts: int):
    """parts + 1 row indices that balance a per-row cost (an int64 tensor of n_rows entries)."""
    n_rows = int(cost.numel())
    pre = torch.zeros(n_rows + 1, dtype=torch.int64, device=cost.device)
    if n_rows:
        pre[1:] = torch.cumsum(cost.to(torch.int64), 0)
    total = int(pre[-1])
    targets = torch.tensor([total * p // parts for p in range(1, parts)], dtype=torch.int64, device=cost.device)
    cuts = torch.searchsorted(pre, targets, right=False).clamp_(0, n_rows)
    bounds = [0] + [int(x) for x in cuts.tolist()] + [n_rows]
    for k in range(1, len(bounds)):
        bounds[k] = max(bounds[k], bounds[k - 1])
    return bounds


def _row_sums(Ap: torch.Tensor, per_entry: torch.Tensor) -> torch.Tensor:
    """sum of per_entry over every CSR row (int64)"""
    pre = torch.zeros(per_entry.numel() + 1, dtype=torch.int64, device=per_entry.device)
    if per_entry.numel():
        pre[1:] = torch.cumsum(per_entry.to(torch.int64), 0)
    ap = Ap.to(torch.int64)
    return pre[ap[1:]] - pre[ap[:-1]]


def mxm_row_cost(Ap, Aj, Bp) -> torch.Tensor:
    """products of every row of A x B: sum over the row's entries (i, k) of the length of B's row k"""
    deg_b = (Bp[1:] - Bp[:-1]).to(torch.int64)
    return _row_sums(Ap, deg_b[Aj.to(torch.int64)])


def mxmT_row_cost(Mp, Mj, Ap, Bp) -> torch.Tensor:
    """merge steps of every row of mask .* (A x B^T): per mask entry (i, j) the lengths of A's row i and B's row j"""
    deg_a = (Ap[1:] - Ap[:-1]).to(torch.int64)
    deg_b = (Bp[1:] - Bp[:-1]).to(torch.int64)
    deg_m = (Mp[1:] - Mp[:-1]).to(torch.int64)
    return _row_sums(Mp, deg_b[Mj.to(torch.int64)]) + deg_a * deg_m


class CsrRows:
    """rows [lo, hi) of a CSR triple as a CSR triple of its own: the row pointers rebased, the column / value arrays VIEWS of the caller's
    (a row block of a matrix every rank holds; a rank that loads only its rows passes them with lo = 0)"""

    def __init__(self, A, lo: int, hi: int):
        Ap, Aj, Ax = A
        s, e = int(Ap[lo]), int(Ap[hi])
        self.lo, self.hi, self.n_rows, self.nnz = lo, hi, hi - lo, e - s
        self.Ap = (Ap[lo:hi + 1] - Ap[lo]).contiguous()
        self.Aj, self.Ax = Aj[s:e], Ax[s:e]

    @property
    def triple(self):
        return self.Ap, self.Aj, self.Ax


def _hip_mxm(t, a_rows, b_cols, A, B, op_mult, op_add, init):
    from . import kernels
    return kernels.mxm(t, a_rows, b_cols, A, B, op_mult, op_add, init)[0]


def _hip_mxmT(t, n_rows, M, A, B, op_mult, op_add, op_select, init):
    from . import kernels
    return kernels.mxmT_masked(t, n_rows, M, A, B, op_mult, op_add, op_select, init)


def sharded_mxm(t, a_rows, b_cols, A, B, op_mult, op_add, init, world: int, rank: int, bounds=None, product=None):
    """This rank's rows of R = A x B.  A, B: CSR triples every rank holds; returns (bounds, (Rp, Rj, Rx) of rows bounds[rank] .. bounds[rank + 1]).
    `product` = the local call (default: the HIP kernels through the C ABI; the CPU tests pass the oracle)."""
    if bounds is None:
        bounds = work_splits(mxm_row_cost(A[0], A[1], B[0]), world)
    rows = CsrRows(A, bounds[rank], bounds[rank + 1])
    return bounds, (product or _hip_mxm)(t, rows.n_rows, b_cols, rows.triple, B, op_mult, op_add, init)


def sharded_mxmT(t, n_rows, M, A, B, op_mult, op_add, op_select, init, world: int, rank: int, bounds=None, product=None):
    """This rank's rows of R = mask .* (A x B^T): the same row block of the mask and of A, B whole."""
    if bounds is None:
        bounds = work_splits(mxmT_row_cost(M[0], M[1], A[0], B[0]), world)
    m, a = CsrRows(M, bounds[rank], bounds[rank + 1]), CsrRows(A, bounds[rank], bounds[rank + 1])
    return bounds, (product or _hip_mxmT)(t, m.n_rows, m.triple, a.triple, B, op_mult, op_add, op_select, init)


def gather_csr_rows(local, bounds, group=None):
    """every rank's row block (Rp, Rj, Rx) -> the whole CSR on every rank (validation; a host read of a sharded result).  One all-gather of the block sizes,
    then one padded all-gather per array."""
    Rp, Rj, Rx = local
    world = len(bounds) - 1
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return Rp, Rj, Rx
    dev = Rp.device
    nnz = torch.tensor([int(Rp[-1])], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, nnz, group=group)
    sizes = [int(s[0]) for s in sizes]
    max_nnz, max_rows = max(sizes + [1]), max(bounds[p + 1] - bounds[p] for p in range(world))

    def gather(x, width, dtype):
        send = torch.zeros(width, dtype=dtype, device=dev)
        send[:x.numel()] = x
        recv = [torch.empty(width, dtype=dtype, device=dev) for _ in range(world)]
        dist.all_gather(recv, send, group=group)
        return recv

    rp = gather(Rp[1:].to(torch.int64), max_rows, torch.int64)
    rj, rx = gather(Rj, max_nnz, Rj.dtype), gather(Rx, max_nnz, Rx.dtype)
    out_p, base = [torch.zeros(1, dtype=torch.int64, device=dev)], 0
    for p in range(world):
        out_p.append(rp[p][:bounds[p + 1] - bounds[p]] + base)
        base += sizes[p]
    return (torch.cat(out_p).to(Rp.dtype), torch.cat([rj[p][:sizes[p]] for p in range(world)]), torch.cat([rx[p][:sizes[p]] for p in range(world)]))
