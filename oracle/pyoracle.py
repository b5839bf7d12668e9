"""pyoracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy-level drivers for the two CPU checkers:

* ``Oracle``    — oracle/liboracle.so, this repo's plain-C restatement of the reference's sequential algorithms
                  (oracle/oracle.c; kind "port").
* ``Reference`` — oracle/_ref/libspla_ref.so, the GENUINE reference CPU backend compiled by oracle/Makefile from
                  /root/reference (kind "reference"), driven through the reference's own C ABI (include/spla.h of
                  the reference) plus the few helpers of oracle/ref_shim.cpp.

Both expose the same four calls over plain CSR / dense / COO numpy arrays of 4-byte words, so a test can check
HIP == Oracle == Reference on the same inputs.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product package (spla_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "liboracle.so"
REF_SO = HERE / "_ref" / "libspla_ref.so"

TYPE = {"INT": 0, "UINT": 1, "FLOAT": 2}
TYPE_NAME = {0: "INT", 1: "UINT", 2: "FLOAT"}
TYPE_NP = {0: np.int32, 1: np.uint32, 2: np.float32}
BINOPS = ["PLUS", "MINUS", "MULT", "DIV", "MINUS_POW2", "FIRST", "SECOND", "BONE", "MIN", "MAX", "LOR", "LAND",
          "BOR", "BAND", "BXOR"]
UNOPS = ["IDENTITY", "AINV", "MINV", "LNOT", "UONE", "ABS", "BNOT", "SQRT", "LOG", "EXP", "SIN", "COS", "TAN", "ASIN", "ACOS", "ATAN",
         "CEIL", "FLOOR", "ROUND", "TRUNC"]
SELOPS = ["EQZERO", "NQZERO", "GTZERO", "GEZERO", "LTZERO", "LEZERO", "ALWAYS", "NEVER"]

_u32p = C.POINTER(C.c_uint32)


def build(force: bool = False) -> None:
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref/libspla_ref.so)."""
    if force or not ORACLE_SO.exists() or ORACLE_SO.stat().st_mtime < (HERE / "oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "port"], check=True, capture_output=True)
    if Path("/root/reference/src/exec.cpp").exists() and (force or not REF_SO.exists()):
        subprocess.run(["make", "-C", str(HERE), "-j8", "ref"], check=True, capture_output=True)
    # the reference's own hot-path unit tests, linked to the product library (make decides what is out of date)
    product = HERE.parent / "spla_amd" / "lib" / "libspla_amd.so"
    if Path("/root/reference/tests/test_mxv.cpp").exists() and product.exists():
        subprocess.run(["make", "-C", str(HERE), "-j4", "reftests", "refexamples"], check=True, capture_output=True)
    # this repo's own C++ tests of the operator API (tests/cpp), same place, same linkage
    if product.exists() and (HERE.parent / "tests" / "cpp").is_dir():
        subprocess.run(["make", "-C", str(HERE), "usertests"], check=True, capture_output=True)


def _tid(t) -> int:
    return TYPE[t.upper()] if isinstance(t, str) else int(t)


def _bid(op) -> int:
    return BINOPS.index(op.upper()) if isinstance(op, str) else int(op)


def _uid(op) -> int:
    return UNOPS.index(op.upper()) if isinstance(op, str) else int(op)


def _sid(op) -> int:
    return SELOPS.index(op.upper()) if isinstance(op, str) else int(op)


def words(a, t=None) -> np.ndarray:
    """View/convert an array as contiguous uint32 words (values of element type t keep their bit pattern)."""
    a = np.asarray(a)
    if t is not None:
        a = a.astype(TYPE_NP[_tid(t)], copy=False)
    assert a.dtype.itemsize == 4, a.dtype
    return np.ascontiguousarray(a).view(np.uint32)


def bits(value, t) -> int:
    return int(np.array([value]).astype(TYPE_NP[_tid(t)]).view(np.uint32)[0])


def _p(a: np.ndarray):
    return a.ctypes.data_as(_u32p)


class Oracle:
    """Plain-C restatement (oracle/oracle.c)."""

    kind = "port"

    def __init__(self):
        build()
        lib = C.CDLL(str(ORACLE_SO))
        lib.oracle_binop.restype = C.c_uint32
        lib.oracle_binop.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32]
        lib.oracle_selop.restype = C.c_int
        lib.oracle_selop.argtypes = [C.c_int, C.c_int, C.c_uint32]
        lib.oracle_mxv_masked.restype = None
        lib.oracle_mxv_masked.argtypes = [C.c_int, C.c_uint32] + [_u32p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int]
        lib.oracle_vxm_masked.restype = C.c_uint32
        lib.oracle_vxm_masked.argtypes = [C.c_int, C.c_uint32, C.c_uint32] + [_u32p] * 4 + [C.c_uint32] + [_u32p] * 4 + [C.c_int] * 3
        lib.oracle_mxm.restype = None
        lib.oracle_mxm.argtypes = [C.c_int, C.c_uint32, C.c_uint32] + [_u32p] * 7 + [C.POINTER(_u32p)] * 2 + [C.c_int, C.c_int, C.c_uint32]
        lib.oracle_mxmT_masked.restype = None
        lib.oracle_mxmT_masked.argtypes = [C.c_int, C.c_uint32] + [_u32p] * 10 + [C.POINTER(_u32p)] * 2 + [C.c_int, C.c_int, C.c_int, C.c_uint32]
        lib.oracle_free.restype = None
        lib.oracle_free.argtypes = [C.c_void_p]
        lib.oracle_v_assign_dense.restype = None
        lib.oracle_v_assign_dense.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, C.c_uint32, C.c_int, C.c_int]
        lib.oracle_v_assign_sparse.restype = None
        lib.oracle_v_assign_sparse.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, C.c_int, C.c_int]
        lib.oracle_v_eadd_fdb_dense.restype = None
        lib.oracle_v_eadd_fdb_dense.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, C.c_int]
        lib.oracle_v_eadd_fdb_sparse.restype = C.c_uint32
        lib.oracle_v_eadd_fdb_sparse.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p, C.c_int]
        lib.oracle_v_count_mf_dense.restype = C.c_uint32
        lib.oracle_v_count_mf_dense.argtypes = [C.c_int, C.c_uint32, _u32p, C.c_uint32]
        lib.oracle_v_eadd_dense.restype = None
        lib.oracle_v_eadd_dense.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, _u32p, C.c_int]
        lib.oracle_v_eadd_sparse.restype = C.c_uint32
        lib.oracle_v_eadd_sparse.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, C.c_uint32, C.c_uint32, _u32p, _u32p, C.c_uint32, _u32p, _u32p, C.c_int]
        lib.oracle_reduce.restype = C.c_uint32
        lib.oracle_reduce.argtypes = [C.c_int, C.c_uint32, _u32p, C.c_uint32, C.c_int]
        lib.oracle_unop.restype = C.c_uint32
        lib.oracle_unop.argtypes = [C.c_int, C.c_int, C.c_uint32]
        lib.oracle_m_transpose.restype = None
        lib.oracle_m_transpose.argtypes = [C.c_int, C.c_uint32, C.c_uint32] + [_u32p] * 6 + [C.c_int]
        lib.oracle_map.restype = None
        lib.oracle_map.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, C.c_int]
        self.lib = lib

    # ---- vector operations of a traversal step: inputs are numpy arrays, r is returned as a new array
    def v_assign_masked(self, t, r, mask, value, op_assign, op_select, mask_idx=None):
        """dense mask (mask = n values) or sparse mask (mask_idx given: mask = the stored values)."""
        t = _tid(t)
        r = words(r, t).copy()
        if mask_idx is None:
            self.lib.oracle_v_assign_dense(t, len(r), _p(r), _p(words(mask, t)), bits(value, t), _bid(op_assign), _sid(op_select))
        else:
            mi = words(mask_idx)
            self.lib.oracle_v_assign_sparse(t, len(mi), _p(r), _p(mi), _p(words(mask, t)), bits(value, t), _bid(op_assign), _sid(op_select))
        return r.view(TYPE_NP[t])

    def v_eadd_fdb(self, t, r, v, op, fdb_fill=0, v_idx=None):
        """returns (r_new, fdb): fdb is a dense array (dense v) or an (indices, values) pair (sparse v: v_idx given)."""
        t = _tid(t)
        r = words(r, t).copy()
        if v_idx is None:
            fdb = np.empty(len(r), np.uint32)
            self.lib.oracle_v_eadd_fdb_dense(t, len(r), _p(r), _p(words(v, t)), _p(fdb), bits(fdb_fill, t), _bid(op))
            return r.view(TYPE_NP[t]), fdb.view(TYPE_NP[t])
        vi = words(v_idx)
        fi, fx = np.empty(max(len(vi), 1), np.uint32), np.empty(max(len(vi), 1), np.uint32)
        n = self.lib.oracle_v_eadd_fdb_sparse(t, len(vi), _p(r), _p(vi), _p(words(v, t)), _p(fi), _p(fx), _bid(op))
        return r.view(TYPE_NP[t]), (fi[:n].copy(), fx[:n].copy().view(TYPE_NP[t]))

    def v_count_mf(self, t, v, fill=0, idx=None, n=None):
        """dense vector: entries != fill; sparse vector (idx given): its stored count, explicit zeros included
        (reference src/cpu/cpu_v_count_mf.hpp:77,88)."""
        t = _tid(t)
        if idx is not None:
            return len(idx)
        v = words(v, t)
        return int(self.lib.oracle_v_count_mf_dense(t, len(v), _p(v), bits(fill, t)))

    # ---- PageRank step remainder / triangle-count tail
    def v_eadd(self, t, u, v, op, u_idx=None, v_idx=None, u_fill=0, v_fill=0):
        """dense x dense (u, v = n values) -> n values; sparse x sparse (u_idx, v_idx given, both index-sorted) -> (indices, values)."""
        t = _tid(t)
        if u_idx is None:
            u, v = words(u, t), words(v, t)
            r = np.empty(len(u), np.uint32)
            self.lib.oracle_v_eadd_dense(t, len(u), _p(r), _p(u), _p(v), _bid(op))
            return r.view(TYPE_NP[t])
        ui, vi = words(u_idx), words(v_idx)
        cap = max(len(ui) + len(vi), 1)
        ri, rx = np.empty(cap, np.uint32), np.empty(cap, np.uint32)
        n = self.lib.oracle_v_eadd_sparse(t, len(ui), _p(ui), _p(words(u, t)), bits(u_fill, t), len(vi), _p(vi), _p(words(v, t)), bits(v_fill, t),
                                          _p(ri), _p(rx), _bid(op))
        return ri[:n].copy(), rx[:n].copy().view(TYPE_NP[t])

    def reduce(self, t, x, op, s=0):
        """strict left fold of x from s: exec_v_reduce over a vector's stored values, exec_m_reduce over a CSR's values."""
        t = _tid(t)
        x = words(x, t)
        out = self.lib.oracle_reduce(t, len(x), _p(x), bits(s, t), _bid(op))
        return np.array([out], np.uint32).view(TYPE_NP[t])[0]

    def m_transpose(self, t, n_rows, n_cols, Ap, Aj, Ax, op_apply="IDENTITY"):
        """R = op(M)^T as CSR (Rp, Rj, Rx)."""
        t = _tid(t)
        Ap, Aj, Ax = words(Ap), words(Aj), words(Ax, t)
        nnz = len(Aj)
        Rp, Rj, Rx = np.zeros(n_cols + 1, np.uint32), np.zeros(max(nnz, 1), np.uint32), np.zeros(max(nnz, 1), np.uint32)
        self.lib.oracle_m_transpose(t, n_rows, n_cols, _p(Ap), _p(Aj if nnz else Rj), _p(Ax if nnz else Rx), _p(Rp), _p(Rj), _p(Rx), _uid(op_apply))
        return Rp, Rj[:nnz], Rx[:nnz].view(TYPE_NP[t])

    def v_map(self, t, x, op):
        """out[i] = op(x[i]): x is the dense image or the stored values of a sparse vector."""
        t = _tid(t)
        x = words(x, t)
        out = np.zeros(max(len(x), 1), np.uint32)
        self.lib.oracle_map(t, len(x), _p(x if len(x) else out), _p(out), _uid(op))
        return out[:len(x)].view(TYPE_NP[t])

    # ---- the graph algorithms, restated over the oracle's own operations (reference src/algorithm.cpp:45-120,158-229,278-335,380-414)
    def bfs(self, n, Ap, Aj, s, pull=False):
        """levels from 1 at the source, 0 = unreached.  push walks rows of the frontier (edge i -> j at A[i, j]); pull tests, per
        unvisited row i, whether any A[i, j] has j in the frontier (the reference's mxv form: edge j -> i)."""
        Ap, Aj = np.asarray(Ap, np.int64), np.asarray(Aj, np.int64)
        depth = np.zeros(n, np.int32)
        front = np.zeros(n, bool)
        front[s] = True
        level = 1
        rows = np.repeat(np.arange(n), np.diff(Ap))
        while front.any():
            depth[front] = level
            if pull:
                hit = np.zeros(n, bool)
                hit[rows[front[Aj]]] = True
            else:
                hit = np.zeros(n, bool)
                hit[Aj[front[rows]]] = True
            front = hit & (depth == 0)
            level += 1
        return depth

    def sssp(self, n, Ap, Aj, Ax, s):
        """fp32 Bellman-Ford to the fixed point, unreachable = FLT_MAX; every candidate is fl(dist[i] + w) as in the reference's min-plus."""
        Ap, Aj, Ax = np.asarray(Ap, np.int64), np.asarray(Aj, np.int64), np.asarray(Ax, np.float32)
        inf = np.float32(np.finfo(np.float32).max)
        dist = np.full(n, inf, np.float32)
        dist[s] = 0
        rows = np.repeat(np.arange(n), np.diff(Ap))
        changed = np.zeros(n, bool)
        changed[s] = True
        while changed.any():
            act = changed[rows]
            with np.errstate(over="ignore"):
                cand = (dist[rows[act]] + Ax[act]).astype(np.float32)
            relaxed = np.full(n, inf, np.float32)
            np.minimum.at(relaxed, Aj[act], cand)
            new = np.minimum(dist, relaxed)
            changed = new != dist
            dist = new
        return dist

    def pr(self, n, Ap, Aj, Ax, alpha, eps):
        """the reference's power iteration step for step, every step through the oracle's own operations"""
        zeros = np.zeros(n, np.float32)
        prev = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
        tele = np.full(n, (np.float32(1.0) - np.float32(alpha)) / np.float32(n), np.float32)
        p = zeros
        err = np.float32(eps) + np.float32(0.1)
        while err > np.float32(eps):
            prod = self.mxv_masked("FLOAT", n, n, Ap, Aj, Ax, zeros, prev, "MULT", "PLUS", "ALWAYS", 0.0)
            p = self.v_eadd("FLOAT", prod, tele, "PLUS")
            d2 = self.v_eadd("FLOAT", p, prev, "MINUS_POW2")
            err = np.sqrt(self.reduce("FLOAT", d2, "PLUS", 0.0))
            p, prev = prev, p
        return prev

    def tc(self, n, Lp, Lj):
        ones = np.ones(len(Lj), np.int32)
        L = (Lp, Lj, ones)
        Bp, Bj, Bx = self.mxmT_masked("INT", n, L, L, L, "MULT", "PLUS", "GTZERO", 0)
        return int(self.reduce("INT", Bx, "PLUS", 0)), (Bp, Bj, Bx)

    def binop(self, t, op, a, b):
        t = _tid(t)
        return np.array([self.lib.oracle_binop(t, _bid(op), bits(a, t), bits(b, t))], dtype=np.uint32).view(TYPE_NP[t])[0]

    def mxv_masked(self, t, n_rows, n_cols, Ap, Aj, Ax, mask, v, op_mult, op_add, op_select, init, early_exit=False):
        t = _tid(t)
        Ap, Aj, Ax, mask, v = words(Ap), words(Aj), words(Ax, t), words(mask, t), words(v, t)
        assert len(Ap) == n_rows + 1 and len(mask) == n_rows and len(v) == n_cols
        r = np.empty(n_rows, dtype=np.uint32)
        self.lib.oracle_mxv_masked(t, n_rows, _p(Ap), _p(Aj), _p(Ax), _p(mask), _p(v), _p(r), _bid(op_mult),
                                   _bid(op_add), _sid(op_select), bits(init, t), 1 if early_exit else 0)
        return r.view(TYPE_NP[t])

    def vxm_masked(self, t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, op_mult, op_add, op_select, init=0):
        t = _tid(t)
        Ap, Aj, Ax, mask, vi, vx = words(Ap), words(Aj), words(Ax, t), words(mask, t), words(vi), words(vx, t)
        assert len(mask) == n_cols and len(vi) == len(vx)
        ri = np.empty(max(n_cols, 1), dtype=np.uint32)
        rx = np.empty(max(n_cols, 1), dtype=np.uint32)
        n = self.lib.oracle_vxm_masked(t, n_rows, n_cols, _p(Ap), _p(Aj), _p(Ax), _p(mask), len(vi), _p(vi), _p(vx),
                                       _p(ri), _p(rx), _bid(op_mult), _bid(op_add), _sid(op_select))
        return ri[:n].copy(), rx[:n].copy().view(TYPE_NP[t])

    def _take(self, Rp, pj, px, t):
        n = int(Rp[-1])
        if n:
            Rj = np.ctypeslib.as_array(pj, shape=(n,)).copy()
            Rx = np.ctypeslib.as_array(px, shape=(n,)).copy().view(TYPE_NP[t])
        else:
            Rj, Rx = np.empty(0, np.uint32), np.empty(0, TYPE_NP[t])
        self.lib.oracle_free(pj)
        self.lib.oracle_free(px)
        return Rp, Rj, Rx

    def mxm(self, t, a_rows, b_cols, A, B, op_mult, op_add, init):
        """A, B: (Ap, Aj, Ax) CSR triples.  Returns the CSR triple of R."""
        t = _tid(t)
        Ap, Aj, Ax = words(A[0]), words(A[1]), words(A[2], t)
        Bp, Bj, Bx = words(B[0]), words(B[1]), words(B[2], t)
        Rp = np.empty(a_rows + 1, dtype=np.uint32)
        pj, px = _u32p(), _u32p()
        self.lib.oracle_mxm(t, a_rows, b_cols, _p(Ap), _p(Aj), _p(Ax), _p(Bp), _p(Bj), _p(Bx), _p(Rp), C.byref(pj),
                            C.byref(px), _bid(op_mult), _bid(op_add), bits(init, t))
        return self._take(Rp, pj, px, t)

    def mxmT_masked(self, t, n_rows, M, A, B, op_mult, op_add, op_select, init):
        t = _tid(t)
        Mp, Mj, Mx = words(M[0]), words(M[1]), words(M[2], t)
        Ap, Aj, Ax = words(A[0]), words(A[1]), words(A[2], t)
        Bp, Bj, Bx = words(B[0]), words(B[1]), words(B[2], t)
        Rp = np.empty(n_rows + 1, dtype=np.uint32)
        pj, px = _u32p(), _u32p()
        self.lib.oracle_mxmT_masked(t, n_rows, _p(Mp), _p(Mj), _p(Mx), _p(Ap), _p(Aj), _p(Ax), _p(Bp), _p(Bj), _p(Bx),
                                    _p(Rp), C.byref(pj), C.byref(px), _bid(op_mult), _bid(op_add), _sid(op_select),
                                    bits(init, t))
        return self._take(Rp, pj, px, t)


def have_reference() -> bool:
    return REF_SO.exists()


class Reference:
    """The genuine reference CPU backend (oracle/_ref/libspla_ref.so) behind the same numpy-level calls."""

    kind = "reference"

    def __init__(self):
        build()
        if not REF_SO.exists():
            raise FileNotFoundError(f"{REF_SO} missing (built only where /root/reference exists)")
        lib = C.CDLL(str(REF_SO))
        self.lib = lib
        vp = C.c_void_p
        for name in ("spla_Type_INT", "spla_Type_UINT", "spla_Type_FLOAT"):
            getattr(lib, name).restype = vp
        lib.refshim_descriptor_make.restype = vp
        lib.refshim_descriptor_make.argtypes = [C.c_int, C.c_int]
        lib.refshim_matrix_lil_count.restype = C.c_uint64
        lib.refshim_matrix_lil_count.argtypes = [vp]
        lib.refshim_matrix_lil_read.argtypes = [vp, _u32p, _u32p, _u32p]
        lib.refshim_vector_dense_read.argtypes = [vp, _u32p]
        lib.refshim_vector_dense_write.argtypes = [vp, _u32p]
        lib.spla_RefCnt_unref.argtypes = [vp]
        lib.spla_MemView_make.argtypes = [C.POINTER(vp), vp, C.c_size_t, C.c_int]
        lib.spla_MemView_get_buffer.argtypes = [vp, C.POINTER(vp)]
        lib.spla_MemView_get_size.argtypes = [vp, C.POINTER(C.c_size_t)]
        lib.spla_Scalar_make.argtypes = [C.POINTER(vp), vp]
        lib.spla_Scalar_set_int.argtypes = [vp, C.c_int]
        lib.spla_Scalar_set_uint.argtypes = [vp, C.c_uint]
        lib.spla_Scalar_set_float.argtypes = [vp, C.c_float]
        lib.spla_Vector_make.argtypes = [C.POINTER(vp), C.c_uint, vp]
        lib.spla_Vector_build.argtypes = [vp, vp, vp]
        lib.spla_Vector_read.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        lib.spla_Matrix_make.argtypes = [C.POINTER(vp), C.c_uint, C.c_uint, vp]
        lib.spla_Matrix_build.argtypes = [vp, vp, vp, vp]
        lib.spla_Exec_mxv_masked.argtypes = [vp] * 10
        lib.spla_Exec_vxm_masked.argtypes = [vp] * 10
        lib.spla_Exec_mxm.argtypes = [vp] * 8
        lib.spla_Exec_mxmT_masked.argtypes = [vp] * 10
        lib.refshim_force_cpu()
        self._types = {0: lib.spla_Type_INT(), 1: lib.spla_Type_UINT(), 2: lib.spla_Type_FLOAT()}

    # ---- object helpers ---------------------------------------------------------------------------------------
    def _ok(self, status, what):
        if status != 0:
            raise RuntimeError(f"reference call {what} returned status {status}")

    def _op(self, kind, name, t):
        fn = getattr(self.lib, f"spla_Op{kind}_{name}_{TYPE_NAME[t]}")
        fn.restype = C.c_void_p
        return fn()

    def _scalar(self, value, t):
        h = C.c_void_p()
        self._ok(self.lib.spla_Scalar_make(C.byref(h), self._types[t]), "Scalar_make")
        if t == 0:
            self.lib.spla_Scalar_set_int(h, int(np.int32(value)))
        elif t == 1:
            self.lib.spla_Scalar_set_uint(h, int(np.uint32(value)))
        else:
            self.lib.spla_Scalar_set_float(h, float(np.float32(value)))
        return h

    def _view(self, arr: np.ndarray):
        h = C.c_void_p()
        self._ok(self.lib.spla_MemView_make(C.byref(h), arr.ctypes.data_as(C.c_void_p), arr.nbytes, 0), "MemView_make")
        return h

    def _unref(self, *handles):
        for h in handles:
            if h:
                self.lib.spla_RefCnt_unref(h)

    def _matrix(self, t, n_rows, n_cols, Ap, Aj, Ax):
        h = C.c_void_p()
        self._ok(self.lib.spla_Matrix_make(C.byref(h), n_rows, n_cols, self._types[t]), "Matrix_make")
        Ap = words(Ap)
        Ai = np.repeat(np.arange(n_rows, dtype=np.uint32), np.diff(Ap.astype(np.int64)))
        Aj, Ax = words(Aj), words(Ax, t)
        v1, v2, v3 = self._view(Ai), self._view(Aj), self._view(Ax)
        self._ok(self.lib.spla_Matrix_build(h, v1, v2, v3), "Matrix_build")
        self._unref(v1, v2, v3)
        return h

    def _dense_vector(self, t, data):
        data = words(data, t)
        h = C.c_void_p()
        self._ok(self.lib.spla_Vector_make(C.byref(h), len(data), self._types[t]), "Vector_make")
        self._ok(self.lib.refshim_vector_dense_write(h, _p(data)), "dense_write")
        return h

    def _empty_vector(self, t, n):
        h = C.c_void_p()
        self._ok(self.lib.spla_Vector_make(C.byref(h), n, self._types[t]), "Vector_make")
        return h

    def _read_lil(self, h, n_rows, t):
        n = int(self.lib.refshim_matrix_lil_count(h))
        Rp = np.empty(n_rows + 1, np.uint32)
        Rj = np.empty(max(n, 1), np.uint32)
        Rx = np.empty(max(n, 1), np.uint32)
        self._ok(self.lib.refshim_matrix_lil_read(h, _p(Rp), _p(Rj), _p(Rx)), "lil_read")
        return Rp, Rj[:n].copy(), Rx[:n].copy().view(TYPE_NP[t])

    # ---- the vector operations of a traversal step (through the reference's own C ABI) -----------------------------
    def _sparse_vector(self, t, n, idx, val, fill=None):
        h = self._empty_vector(t, n)
        if fill is not None:  # before build: set_fill_value drops the content (reference src/core/tvector.hpp:131-140)
            self.lib.spla_Vector_set_fill_value.argtypes = [C.c_void_p, C.c_void_p]
            hs = self._scalar(fill, t)
            self.lib.spla_Vector_set_fill_value(h, hs)
            self._unref(hs)
        idx, val = words(idx), words(val, t)
        k1, k2 = self._view(idx), self._view(val)
        self._ok(self.lib.spla_Vector_build(h, k1, k2), "Vector_build")
        self._unref(k1, k2)
        return h

    def _read_coo(self, h, t):
        rk, rv = C.c_void_p(), C.c_void_p()
        self._ok(self.lib.spla_Vector_read(h, C.byref(rk), C.byref(rv)), "Vector_read")
        buf, size = C.c_void_p(), C.c_size_t()
        self.lib.spla_MemView_get_buffer(rk, C.byref(buf))
        self.lib.spla_MemView_get_size(rk, C.byref(size))
        n = size.value // 4
        ri = np.ctypeslib.as_array(C.cast(buf, _u32p), shape=(n,)).copy() if n else np.empty(0, np.uint32)
        self.lib.spla_MemView_get_buffer(rv, C.byref(buf))
        rx = np.ctypeslib.as_array(C.cast(buf, _u32p), shape=(n,)).copy() if n else np.empty(0, np.uint32)
        self._unref(rk, rv)
        return ri, rx.view(TYPE_NP[t])

    def _read_dense(self, h, n, t):
        out = np.empty(n, np.uint32)
        self._ok(self.lib.refshim_vector_dense_read(h, _p(out)), "dense_read")
        return out.view(TYPE_NP[t])

    def v_assign_masked(self, t, r, mask, value, op_assign, op_select, mask_idx=None):
        t = _tid(t)
        n = len(r)
        self.lib.spla_Exec_v_assign_masked.argtypes = [C.c_void_p] * 7
        hr = self._dense_vector(t, r)
        hm = self._dense_vector(t, mask) if mask_idx is None else self._sparse_vector(t, n, mask_idx, mask)
        hv = self._scalar(value, t)
        self._ok(self.lib.spla_Exec_v_assign_masked(hr, hm, hv, self._op("Binary", BINOPS[_bid(op_assign)], t),
                                                    self._op("Select", SELOPS[_sid(op_select)], t), None, None), "Exec_v_assign_masked")
        out = self._read_dense(hr, n, t)
        self._unref(hr, hm, hv)
        return out

    def v_eadd_fdb(self, t, r, v, op, fdb_fill=0, v_idx=None):
        t = _tid(t)
        n = len(r)
        self.lib.spla_Exec_v_eadd_fdb.argtypes = [C.c_void_p] * 6
        self.lib.spla_Vector_set_fill_value.argtypes = [C.c_void_p, C.c_void_p]
        hr = self._dense_vector(t, r)
        hv = self._dense_vector(t, v) if v_idx is None else self._sparse_vector(t, n, v_idx, v)
        hf = self._empty_vector(t, n)
        hs = self._scalar(fdb_fill, t)
        self.lib.spla_Vector_set_fill_value(hf, hs)
        self._ok(self.lib.spla_Exec_v_eadd_fdb(hr, hv, hf, self._op("Binary", BINOPS[_bid(op)], t), None, None), "Exec_v_eadd_fdb")
        r_new = self._read_dense(hr, n, t)
        fdb = self._read_dense(hf, n, t) if v_idx is None else self._read_coo(hf, t)
        self._unref(hr, hv, hf, hs)
        return r_new, fdb

    def v_count_mf(self, t, v, fill=0, idx=None, n=None):
        """dense vector (v = n values) or sparse (idx given): the reference answers with the stored count for sparse formats."""
        t = _tid(t)
        self.lib.spla_Exec_v_count_mf.argtypes = [C.c_void_p] * 4
        self.lib.spla_Scalar_get_uint.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        self.lib.spla_Vector_set_fill_value.argtypes = [C.c_void_p, C.c_void_p]
        if idx is None:
            hv = self._empty_vector(t, len(v))
            hs = self._scalar(fill, t)
            self.lib.spla_Vector_set_fill_value(hv, hs)
            self._ok(self.lib.refshim_vector_dense_write(hv, _p(words(v, t))), "dense_write")
            self._unref(hs)
        else:
            hv = self._sparse_vector(t, n, idx, v)
        hr = self._scalar(0, 1)
        self._ok(self.lib.spla_Exec_v_count_mf(hr, hv, None, None), "Exec_v_count_mf")
        out = C.c_uint()
        self.lib.spla_Scalar_get_uint(hr, C.byref(out))
        self._unref(hr, hv)
        return int(out.value)

    # ---- PageRank step remainder / triangle-count tail and the graph algorithms (through the reference's own C ABI)
    def _scalar_value(self, h, t):
        if t == 0:
            out = C.c_int()
            self.lib.spla_Scalar_get_int.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            self.lib.spla_Scalar_get_int(h, C.byref(out))
            return np.int32(out.value)
        if t == 1:
            out = C.c_uint()
            self.lib.spla_Scalar_get_uint.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
            self.lib.spla_Scalar_get_uint(h, C.byref(out))
            return np.uint32(out.value)
        out = C.c_float()
        self.lib.spla_Scalar_get_float.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self.lib.spla_Scalar_get_float(h, C.byref(out))
        return np.float32(out.value)

    def v_eadd(self, t, u, v, op, u_idx=None, v_idx=None, u_fill=0, v_fill=0):
        t = _tid(t)
        self.lib.spla_Exec_v_eadd.argtypes = [C.c_void_p] * 6
        self.lib.spla_Vector_set_fill_value.argtypes = [C.c_void_p, C.c_void_p]
        if u_idx is None:
            n = len(u)
            hu, hv = self._dense_vector(t, u), self._dense_vector(t, v)
        else:
            n = int(max(list(u_idx) + list(v_idx) + [0])) + 1
            hu, hv = self._sparse_vector(t, n, u_idx, u, fill=u_fill), self._sparse_vector(t, n, v_idx, v, fill=v_fill)
        hr = self._empty_vector(t, n)
        self._ok(self.lib.spla_Exec_v_eadd(hr, hu, hv, self._op("Binary", BINOPS[_bid(op)], t), None, None), "Exec_v_eadd")
        out = self._read_dense(hr, n, t) if u_idx is None else self._read_coo(hr, t)
        self._unref(hr, hu, hv)
        return out

    def reduce(self, t, x, op, s=0, idx=None, n=None):
        """exec_v_reduce over a dense vector (x = n values) or a sparse one (idx given: x = its stored values)."""
        t = _tid(t)
        self.lib.spla_Exec_v_reduce.argtypes = [C.c_void_p] * 6
        hv = self._dense_vector(t, x) if idx is None else self._sparse_vector(t, n, idx, x)
        hs, hr = self._scalar(s, t), self._scalar(0, t)
        self._ok(self.lib.spla_Exec_v_reduce(hr, hs, hv, self._op("Binary", BINOPS[_bid(op)], t), None, None), "Exec_v_reduce")
        out = self._scalar_value(hr, t)
        self._unref(hv, hs, hr)
        return out

    def m_reduce(self, t, n_rows, n_cols, Ap, Aj, Ax, op, s=0):
        t = _tid(t)
        self.lib.spla_Exec_m_reduce.argtypes = [C.c_void_p] * 6
        M = self._matrix(t, n_rows, n_cols, Ap, Aj, Ax)
        hs, hr = self._scalar(s, t), self._scalar(0, t)
        self._ok(self.lib.spla_Exec_m_reduce(hr, hs, M, self._op("Binary", BINOPS[_bid(op)], t), None, None), "Exec_m_reduce")
        out = self._scalar_value(hr, t)
        self._unref(M, hs, hr)
        return out

    def m_transpose(self, t, n_rows, n_cols, Ap, Aj, Ax, op_apply="IDENTITY"):
        t = _tid(t)
        self.lib.spla_Exec_m_transpose.argtypes = [C.c_void_p] * 5
        M = self._matrix(t, n_rows, n_cols, Ap, Aj, Ax)
        R = C.c_void_p()
        self._ok(self.lib.spla_Matrix_make(C.byref(R), n_cols, n_rows, self._types[t]), "Matrix_make")
        self._ok(self.lib.spla_Exec_m_transpose(R, M, self._op("Unary", UNOPS[_uid(op_apply)], t), None, None), "Exec_m_transpose")
        out = self._read_lil(R, n_cols, t)
        self._unref(M, R)
        return out

    def v_map(self, t, x, op, idx=None, n=None):
        """exec_v_map over a dense vector, or a sparse one (idx given); returns the dense image / (indices, values)."""
        t = _tid(t)
        self.lib.spla_Exec_v_map.argtypes = [C.c_void_p] * 5
        hv = self._dense_vector(t, x) if idx is None else self._sparse_vector(t, n, idx, x)
        hr = self._empty_vector(t, len(x) if idx is None else n)
        self._ok(self.lib.spla_Exec_v_map(hr, hv, self._op("Unary", UNOPS[_uid(op)], t), None, None), "Exec_v_map")
        out = self._read_dense(hr, len(x), t) if idx is None else self._read_coo(hr, t)
        self._unref(hv, hr)
        return out

    def _traversal(self, mode, front_factor):
        self.lib.refshim_descriptor_make_traversal.restype = C.c_void_p
        self.lib.refshim_descriptor_make_traversal.argtypes = [C.c_int, C.c_float]
        return self.lib.refshim_descriptor_make_traversal({"push": 0, "pull": 1, "push_pull": 2}[mode], float(front_factor))

    def bfs(self, n, Ap, Aj, s, mode="push_pull", front_factor=0.1):
        """spla_Algorithm_bfs on the INT adjacency matrix (values 1); returns the n levels."""
        self.lib.spla_Algorithm_bfs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        A = self._matrix(0, n, n, Ap, Aj, np.ones(len(Aj), np.int32))
        hv, d = self._empty_vector(0, n), self._traversal(mode, front_factor)
        self._ok(self.lib.spla_Algorithm_bfs(hv, A, int(s), d), "Algorithm_bfs")
        out = self._read_dense(hv, n, 0)
        self._unref(A, hv, d)
        return out

    def sssp(self, n, Ap, Aj, Ax, s, mode="push_pull", front_factor=0.1):
        self.lib.spla_Algorithm_sssp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        A = self._matrix(2, n, n, Ap, Aj, Ax)
        hv, d = self._empty_vector(2, n), self._traversal(mode, front_factor)
        self._ok(self.lib.spla_Algorithm_sssp(hv, A, int(s), d), "Algorithm_sssp")
        out = self._read_dense(hv, n, 2)
        self._unref(A, hv, d)
        return out

    def pr(self, n, Ap, Aj, Ax, alpha, eps):
        self.lib.spla_Algorithm_pr.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        A = self._matrix(2, n, n, Ap, Aj, Ax)
        hp, d = self._empty_vector(2, n), self._traversal("push_pull", 0.1)
        self._ok(self.lib.spla_Algorithm_pr(C.byref(hp), A, float(alpha), float(eps), d), "Algorithm_pr")
        out = self._read_dense(hp, n, 2)
        self._unref(A, hp, d)
        return out

    def tc(self, n, Lp, Lj):
        """spla_Algorithm_tc on the strict lower triangle L (INT ones): returns (ntrins, (Bp, Bj, Bx))."""
        self.lib.spla_Algorithm_tc.argtypes = [C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
        L = self._matrix(0, n, n, Lp, Lj, np.ones(len(Lj), np.int32))
        B = C.c_void_p()
        self._ok(self.lib.spla_Matrix_make(C.byref(B), n, n, self._types[0]), "Matrix_make")
        d, count = self._traversal("push_pull", 0.1), C.c_int(0)
        self._ok(self.lib.spla_Algorithm_tc(C.byref(count), L, B, d), "Algorithm_tc")
        out = self._read_lil(B, n, 0)
        self._unref(L, B, d)
        return int(count.value), out

    def load_mtx(self, path, offset_indices=True, make_undirected=True, remove_loops=True):
        """the reference's MtxLoader::load -> (n_rows, n_cols, Ai, Aj)"""
        lib = self.lib
        lib.refshim_mtx_load.restype = C.c_void_p
        lib.refshim_mtx_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        lib.refshim_mtx_get.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(_u32p), C.POINTER(_u32p)]
        lib.refshim_mtx_free.argtypes = [C.c_void_p]
        h = lib.refshim_mtx_load(str(path).encode(), int(offset_indices), int(make_undirected), int(remove_loops))
        if not h:
            raise RuntimeError("reference MtxLoader failed")
        nr, nc, nv, pi, pj = C.c_uint32(), C.c_uint32(), C.c_uint64(), _u32p(), _u32p()
        lib.refshim_mtx_get(h, C.byref(nr), C.byref(nc), C.byref(nv), C.byref(pi), C.byref(pj))
        n = nv.value
        Ai = np.ctypeslib.as_array(pi, shape=(n,)).copy() if n else np.empty(0, np.uint32)
        Aj = np.ctypeslib.as_array(pj, shape=(n,)).copy() if n else np.empty(0, np.uint32)
        lib.refshim_mtx_free(h)
        return nr.value, nc.value, Ai, Aj

    # ---- the four operations ----------------------------------------------------------------------------------
    def mxv_masked(self, t, n_rows, n_cols, Ap, Aj, Ax, mask, v, op_mult, op_add, op_select, init, early_exit=False):
        t = _tid(t)
        M = self._matrix(t, n_rows, n_cols, Ap, Aj, Ax)
        hm, hv, hr = self._dense_vector(t, mask), self._dense_vector(t, v), self._empty_vector(t, n_rows)
        hi = self._scalar(init, t)
        desc = self.lib.refshim_descriptor_make(1 if early_exit else 0, 0)
        self._ok(self.lib.spla_Exec_mxv_masked(hr, hm, M, hv, self._op("Binary", BINOPS[_bid(op_mult)], t),
                                               self._op("Binary", BINOPS[_bid(op_add)], t),
                                               self._op("Select", SELOPS[_sid(op_select)], t), hi, desc, None), "Exec_mxv_masked")
        r = np.empty(n_rows, np.uint32)
        self._ok(self.lib.refshim_vector_dense_read(hr, _p(r)), "dense_read")
        self._unref(M, hm, hv, hr, hi, desc)
        return r.view(TYPE_NP[t])

    def vxm_masked(self, t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, op_mult, op_add, op_select, init=0):
        t = _tid(t)
        M = self._matrix(t, n_rows, n_cols, Ap, Aj, Ax)
        hm, hr = self._dense_vector(t, mask), self._empty_vector(t, n_cols)
        hv = self._empty_vector(t, n_rows)
        vi, vx = words(vi), words(vx, t)
        k1, k2 = self._view(vi), self._view(vx)
        self._ok(self.lib.spla_Vector_build(hv, k1, k2), "Vector_build")
        hi = self._scalar(init, t)
        self._ok(self.lib.spla_Exec_vxm_masked(hr, hm, hv, M, self._op("Binary", BINOPS[_bid(op_mult)], t),
                                               self._op("Binary", BINOPS[_bid(op_add)], t),
                                               self._op("Select", SELOPS[_sid(op_select)], t), hi, None, None), "Exec_vxm_masked")
        rk, rv = C.c_void_p(), C.c_void_p()
        self._ok(self.lib.spla_Vector_read(hr, C.byref(rk), C.byref(rv)), "Vector_read")
        buf, size = C.c_void_p(), C.c_size_t()
        self.lib.spla_MemView_get_buffer(rk, C.byref(buf))
        self.lib.spla_MemView_get_size(rk, C.byref(size))
        n = size.value // 4
        ri = np.ctypeslib.as_array(C.cast(buf, _u32p), shape=(n,)).copy() if n else np.empty(0, np.uint32)
        self.lib.spla_MemView_get_buffer(rv, C.byref(buf))
        rx = np.ctypeslib.as_array(C.cast(buf, _u32p), shape=(n,)).copy() if n else np.empty(0, np.uint32)
        self._unref(rk, rv, k1, k2, M, hm, hr, hv, hi)
        return ri, rx.view(TYPE_NP[t])

    def mxm(self, t, a_rows, b_cols, A, B, op_mult, op_add, init):
        t = _tid(t)
        a_cols = len(B[0]) - 1
        hA = self._matrix(t, a_rows, a_cols, *A)
        hB = self._matrix(t, a_cols, b_cols, *B)
        hR = C.c_void_p()
        self._ok(self.lib.spla_Matrix_make(C.byref(hR), a_rows, b_cols, self._types[t]), "Matrix_make")
        hi = self._scalar(init, t)
        self._ok(self.lib.spla_Exec_mxm(hR, hA, hB, self._op("Binary", BINOPS[_bid(op_mult)], t),
                                        self._op("Binary", BINOPS[_bid(op_add)], t), hi, None, None), "Exec_mxm")
        out = self._read_lil(hR, a_rows, t)
        self._unref(hA, hB, hR, hi)
        return out

    def mxmT_masked(self, t, n_rows, M, A, B, op_mult, op_add, op_select, init, n_cols=None, inner=None):
        t = _tid(t)
        b_rows = len(B[0]) - 1
        n_cols = b_rows if n_cols is None else n_cols
        inner = int(max(np.max(A[1], initial=0), np.max(B[1], initial=0))) + 1 if inner is None else inner
        hM = self._matrix(t, n_rows, n_cols, *M)
        hA = self._matrix(t, n_rows, inner, *A)
        hB = self._matrix(t, b_rows, inner, *B)
        hR = C.c_void_p()
        self._ok(self.lib.spla_Matrix_make(C.byref(hR), n_rows, n_cols, self._types[t]), "Matrix_make")
        hi = self._scalar(init, t)
        self._ok(self.lib.spla_Exec_mxmT_masked(hR, hM, hA, hB, self._op("Binary", BINOPS[_bid(op_mult)], t),
                                                self._op("Binary", BINOPS[_bid(op_add)], t),
                                                self._op("Select", SELOPS[_sid(op_select)], t), hi, None, None), "Exec_mxmT_masked")
        out = self._read_lil(hR, n_rows, t)
        self._unref(hM, hA, hB, hR, hi)
        return out
