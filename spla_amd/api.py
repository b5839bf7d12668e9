"""Python face of the operator API: a thin ctypes mirror of the reference's `pyspla` call shape for the hot path
(reference python/pyspla/matrix.py:574-1015 `Matrix.mxm / mxmT / mxv`, vector.py:476-546 `Vector.vxm`,
bridge.py:162-171 the ctypes loading), over this repo's include/spla.h.

    from spla_amd.api import INT, Matrix, Vector
    M = Matrix.from_lists([0, 1, 2], [1, 2, 0], [1, 2, 3], (3, 3), INT)
    v = Vector.from_lists([0, 1, 2], [1, 1, 1], 3, INT)
    r = M.mxv(mask, v, INT.MULT, INT.PLUS, INT.NQZERO)

Every compute call goes through libspla_amd.so (`spla_Exec_*` -> dispatcher -> HIP kernels); there is no Python or
CPU fallback: a non-zero status raises ``SplaError``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .ops import UNOP, unop_valid

_vp = C.c_void_p

STATUS = {0: "OK", 1: "ERROR", 2: "NO_ACCELERATION", 3: "PLATFORM_NOT_FOUND", 4: "DEVICE_NOT_FOUND", 5: "INVALID_STATE",
          6: "INVALID_ARGUMENT", 7: "NO_VALUE", 1024: "NOT_IMPLEMENTED"}
ACCELERATOR_NONE, ACCELERATOR_OPENCL, ACCELERATOR_HIP = 0, 1, 2

BINARY_OPS = ["PLUS", "MINUS", "MULT", "DIV", "MINUS_POW2", "FIRST", "SECOND", "BONE", "MIN", "MAX", "LOR", "LAND", "BOR", "BAND", "BXOR"]
SELECT_OPS = ["EQZERO", "NQZERO", "GTZERO", "GEZERO", "LTZERO", "LEZERO", "ALWAYS", "NEVER"]
BITWISE = ("BOR", "BAND", "BXOR")
UNARY_OPS = list(UNOP)


class SplaError(RuntimeError):
    def __init__(self, status, what):
        super().__init__(f"{what}: status {status} ({STATUS.get(status, '?')})")
        self.status = status


def c_symbols():
    """Every symbol include/spla.h declares (used by the CPU-side export test)."""
    names = ["spla_Library_finalize", "spla_Library_initialize", "spla_Library_set_accelerator", "spla_Library_set_platform",
             "spla_Library_set_device", "spla_Library_set_queues_count", "spla_Library_set_message_callback",
             "spla_Library_set_default_callback", "spla_Library_get_accelerator_info",
             "spla_Type_BOOL", "spla_Type_INT", "spla_Type_UINT", "spla_Type_FLOAT",
             "spla_RefCnt_ref", "spla_RefCnt_unref",
             "spla_MemView_make", "spla_MemView_read", "spla_MemView_write", "spla_MemView_get_buffer", "spla_MemView_get_size", "spla_MemView_is_mutable",
             "spla_Scalar_make", "spla_Scalar_set_int", "spla_Scalar_set_uint", "spla_Scalar_set_float", "spla_Scalar_get_int", "spla_Scalar_get_uint", "spla_Scalar_get_float",
             "spla_Vector_make", "spla_Vector_set_format", "spla_Vector_set_fill_value", "spla_Vector_set_reduce", "spla_Vector_set_int", "spla_Vector_set_uint",
             "spla_Vector_set_float", "spla_Vector_get_int", "spla_Vector_get_uint", "spla_Vector_get_float", "spla_Vector_build", "spla_Vector_read", "spla_Vector_clear",
             "spla_Matrix_make", "spla_Matrix_set_format", "spla_Matrix_set_fill_value", "spla_Matrix_set_reduce", "spla_Matrix_set_int", "spla_Matrix_set_uint",
             "spla_Matrix_set_float", "spla_Matrix_get_int", "spla_Matrix_get_uint", "spla_Matrix_get_float", "spla_Matrix_build", "spla_Matrix_read", "spla_Matrix_clear",
             "spla_Exec_mxm", "spla_Exec_mxmT_masked", "spla_Exec_mxv_masked", "spla_Exec_vxm_masked",
             "spla_Exec_v_eadd_fdb", "spla_Exec_v_assign_masked", "spla_Exec_v_count_mf",
             "spla_Exec_v_eadd", "spla_Exec_v_reduce", "spla_Exec_m_reduce", "spla_Exec_m_transpose", "spla_Exec_v_map",
             "spla_Algorithm_bfs", "spla_Algorithm_sssp", "spla_Algorithm_pr", "spla_Algorithm_tc",
             "spla_MtxLoader_load", "spla_MtxLoader_get", "spla_MtxLoader_delete",
             "spla_Descriptor_make", "spla_Descriptor_set_early_exit", "spla_Descriptor_set_struct_only", "spla_Descriptor_set_deterministic", "spla_Descriptor_set_reference_order",
             "spla_Descriptor_set_traversal_mode", "spla_Descriptor_set_front_factor"]
    for op in BINARY_OPS:
        for t in ("INT", "UINT") + (() if op in BITWISE else ("FLOAT",)):
            names.append(f"spla_OpBinary_{op}_{t}")
    for op in SELECT_OPS:
        for t in ("INT", "UINT", "FLOAT"):
            names.append(f"spla_OpSelect_{op}_{t}")
    for op in UNARY_OPS:
        for t in ("INT", "UINT", "FLOAT"):
            if unop_valid(op, t):
                names.append(f"spla_OpUnary_{op}_{t}")
    return names


_bound = False


def _c():
    global _bound
    lib = _lib.load()
    _lib._sync_env(lib)  # (under the test suite: the operator API sees a changed SPLAHIP_* knob like the kernel-level calls do)
    if not _bound:
        for name in c_symbols():
            fn = getattr(lib, name)
            if name.startswith(("spla_Type_", "spla_OpBinary_", "spla_OpSelect_", "spla_OpUnary_")):
                fn.restype, fn.argtypes = _vp, []
            elif name == "spla_Library_finalize":
                fn.restype, fn.argtypes = None, []
            else:
                fn.restype = C.c_int
        lib.spla_Library_set_accelerator.argtypes = [C.c_int]
        lib.spla_Library_set_device.argtypes = [C.c_int]
        lib.spla_Library_get_accelerator_info.argtypes = [C.c_char_p, C.c_int]
        lib.spla_RefCnt_ref.argtypes = [_vp]
        lib.spla_RefCnt_unref.argtypes = [_vp]
        lib.spla_MemView_make.argtypes = [C.POINTER(_vp), _vp, C.c_size_t, C.c_int]
        lib.spla_MemView_get_buffer.argtypes = [_vp, C.POINTER(_vp)]
        lib.spla_MemView_get_size.argtypes = [_vp, C.POINTER(C.c_size_t)]
        lib.spla_Scalar_make.argtypes = [C.POINTER(_vp), _vp]
        lib.spla_Scalar_set_int.argtypes = [_vp, C.c_int]
        lib.spla_Scalar_set_uint.argtypes = [_vp, C.c_uint]
        lib.spla_Scalar_set_float.argtypes = [_vp, C.c_float]
        lib.spla_Scalar_get_int.argtypes = [_vp, C.POINTER(C.c_int)]
        lib.spla_Scalar_get_uint.argtypes = [_vp, C.POINTER(C.c_uint)]
        lib.spla_Scalar_get_float.argtypes = [_vp, C.POINTER(C.c_float)]
        lib.spla_Vector_make.argtypes = [C.POINTER(_vp), C.c_uint, _vp]
        lib.spla_Vector_set_format.argtypes = [_vp, C.c_int]
        lib.spla_Vector_set_fill_value.argtypes = [_vp, _vp]
        lib.spla_Vector_set_reduce.argtypes = [_vp, _vp]
        lib.spla_Vector_set_int.argtypes = [_vp, C.c_uint, C.c_int]
        lib.spla_Vector_set_uint.argtypes = [_vp, C.c_uint, C.c_uint]
        lib.spla_Vector_set_float.argtypes = [_vp, C.c_uint, C.c_float]
        lib.spla_Vector_get_int.argtypes = [_vp, C.c_uint, C.POINTER(C.c_int)]
        lib.spla_Vector_get_uint.argtypes = [_vp, C.c_uint, C.POINTER(C.c_uint)]
        lib.spla_Vector_get_float.argtypes = [_vp, C.c_uint, C.POINTER(C.c_float)]
        lib.spla_Vector_build.argtypes = [_vp, _vp, _vp]
        lib.spla_Vector_read.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp)]
        lib.spla_Vector_clear.argtypes = [_vp]
        lib.spla_Matrix_make.argtypes = [C.POINTER(_vp), C.c_uint, C.c_uint, _vp]
        lib.spla_Matrix_set_format.argtypes = [_vp, C.c_int]
        lib.spla_Matrix_set_fill_value.argtypes = [_vp, _vp]
        lib.spla_Matrix_set_reduce.argtypes = [_vp, _vp]
        lib.spla_Matrix_set_int.argtypes = [_vp, C.c_uint, C.c_uint, C.c_int]
        lib.spla_Matrix_set_uint.argtypes = [_vp, C.c_uint, C.c_uint, C.c_uint]
        lib.spla_Matrix_set_float.argtypes = [_vp, C.c_uint, C.c_uint, C.c_float]
        lib.spla_Matrix_get_int.argtypes = [_vp, C.c_uint, C.c_uint, C.POINTER(C.c_int)]
        lib.spla_Matrix_get_uint.argtypes = [_vp, C.c_uint, C.c_uint, C.POINTER(C.c_uint)]
        lib.spla_Matrix_get_float.argtypes = [_vp, C.c_uint, C.c_uint, C.POINTER(C.c_float)]
        lib.spla_Matrix_build.argtypes = [_vp, _vp, _vp, _vp]
        lib.spla_Matrix_read.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]
        lib.spla_Matrix_clear.argtypes = [_vp]
        lib.spla_Exec_mxm.argtypes = [_vp] * 7 + [C.POINTER(_vp)]
        lib.spla_Exec_mxmT_masked.argtypes = [_vp] * 9 + [C.POINTER(_vp)]
        lib.spla_Exec_mxv_masked.argtypes = [_vp] * 9 + [C.POINTER(_vp)]
        lib.spla_Exec_vxm_masked.argtypes = [_vp] * 9 + [C.POINTER(_vp)]
        lib.spla_Exec_v_eadd_fdb.argtypes = [_vp] * 5 + [C.POINTER(_vp)]
        lib.spla_Exec_v_assign_masked.argtypes = [_vp] * 6 + [C.POINTER(_vp)]
        lib.spla_Exec_v_count_mf.argtypes = [_vp] * 3 + [C.POINTER(_vp)]
        lib.spla_Exec_v_eadd.argtypes = [_vp] * 5 + [C.POINTER(_vp)]
        lib.spla_Exec_v_reduce.argtypes = [_vp] * 5 + [C.POINTER(_vp)]
        lib.spla_Exec_m_reduce.argtypes = [_vp] * 5 + [C.POINTER(_vp)]
        lib.spla_Exec_m_transpose.argtypes = [_vp] * 4 + [C.POINTER(_vp)]
        lib.spla_Exec_v_map.argtypes = [_vp] * 4 + [C.POINTER(_vp)]
        lib.spla_Algorithm_bfs.argtypes = [_vp, _vp, C.c_uint, _vp]
        lib.spla_Algorithm_sssp.argtypes = [_vp, _vp, C.c_uint, _vp]
        lib.spla_Algorithm_pr.argtypes = [C.POINTER(_vp), _vp, C.c_float, C.c_float, _vp]
        lib.spla_Algorithm_tc.argtypes = [C.POINTER(C.c_int), _vp, _vp, _vp]
        lib.spla_Descriptor_set_traversal_mode.argtypes = [_vp, C.c_int]
        lib.spla_Descriptor_set_front_factor.argtypes = [_vp, C.c_float]
        lib.spla_MtxLoader_load.argtypes = [C.POINTER(_vp), C.c_char_p, C.c_int, C.c_int, C.c_int]
        lib.spla_MtxLoader_get.argtypes = [_vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32))]
        lib.spla_MtxLoader_delete.argtypes = [_vp]
        lib.spla_Descriptor_make.argtypes = [C.POINTER(_vp)]
        lib.spla_Descriptor_set_early_exit.argtypes = [_vp, C.c_int]
        lib.spla_Descriptor_set_struct_only.argtypes = [_vp, C.c_int]
        lib.spla_Descriptor_set_deterministic.argtypes = [_vp, C.c_int]
        lib.spla_Descriptor_set_reference_order.argtypes = [_vp, C.c_int]
        _bound = True
    return lib


def _check(status, what):
    if status != 0:
        raise SplaError(status, what)


# ---- library --------------------------------------------------------------------------------------------------------
def set_accelerator(kind=ACCELERATOR_HIP):
    _check(_c().spla_Library_set_accelerator(kind), "spla_Library_set_accelerator")


def set_device(index):
    _check(_c().spla_Library_set_device(index), "spla_Library_set_device")


def accelerator_info() -> str:
    buf = C.create_string_buffer(512)
    _c().spla_Library_get_accelerator_info(buf, 512)
    return buf.value.decode()


# ---- element types + their ops (pyspla exposes ops as attributes of the type class, python/pyspla/type.py:210-286) --------
class _Op:
    def __init__(self, hnd, name):
        self.hnd, self.name = hnd, name

    def __repr__(self):
        return f"<op {self.name}>"


class Type:
    _name, _np, _code = None, None, None

    @classmethod
    def _hnd(cls):
        return getattr(_c(), f"spla_Type_{cls._name}")()

    @classmethod
    def _op(cls, kind, op):
        fn = getattr(_c(), f"spla_Op{kind}_{op}_{cls._name}")
        return _Op(fn(), f"{op}_{cls._name}")

    @classmethod
    def _scalar_set(cls, s, v):
        lib = _c()
        if cls is INT:
            lib.spla_Scalar_set_int(s, int(np.int32(v)))
        elif cls is UINT:
            lib.spla_Scalar_set_uint(s, int(np.uint32(v)))
        else:
            lib.spla_Scalar_set_float(s, float(np.float32(v)))


class _OpAttr:
    def __init__(self, kind, op):
        self.kind, self.op = kind, op

    def __get__(self, obj, owner):
        return owner._op(self.kind, self.op)


for _op in BINARY_OPS:
    setattr(Type, _op, _OpAttr("Binary", _op))
for _op in SELECT_OPS:
    setattr(Type, _op, _OpAttr("Select", _op))
for _op in UNARY_OPS:  # e.g. INT.AINV, FLOAT.SQRT; an op the type lacks (FLOAT.BNOT, INT.SQRT) raises AttributeError at lookup
    setattr(Type, _op, _OpAttr("Unary", _op))


class INT(Type):
    _name, _np, _code = "INT", np.int32, "I"


class UINT(Type):
    _name, _np, _code = "UINT", np.uint32, "U"


class FLOAT(Type):
    _name, _np, _code = "FLOAT", np.float32, "F"


# ---- objects --------------------------------------------------------------------------------------------------------------
class Object:
    def __init__(self, hnd):
        self.hnd = hnd

    def __del__(self):
        try:
            if self.hnd:
                _c().spla_RefCnt_unref(self.hnd)
                self.hnd = None
        except Exception:
            pass


class MemView(Object):
    def __init__(self, array: np.ndarray = None, hnd=None):
        self._keep = array
        if hnd is None:
            h = _vp()
            _check(_c().spla_MemView_make(C.byref(h), array.ctypes.data_as(_vp), array.nbytes, 0), "spla_MemView_make")
            hnd = h
        super().__init__(hnd)

    def to_numpy(self, dtype):
        buf, size = _vp(), C.c_size_t()
        _c().spla_MemView_get_buffer(self.hnd, C.byref(buf))
        _c().spla_MemView_get_size(self.hnd, C.byref(size))
        n = size.value // 4
        if n == 0:
            return np.empty(0, dtype)
        return np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint32)), shape=(n,)).copy().view(dtype)


class Scalar(Object):
    def __init__(self, dtype=INT, value=None):
        h = _vp()
        _check(_c().spla_Scalar_make(C.byref(h), dtype._hnd()), "spla_Scalar_make")
        super().__init__(h)
        self.dtype = dtype
        if value is not None:
            dtype._scalar_set(self.hnd, value)

    def get(self):
        lib = _c()
        if self.dtype is INT:
            v = C.c_int()
            lib.spla_Scalar_get_int(self.hnd, C.byref(v))
        elif self.dtype is UINT:
            v = C.c_uint()
            lib.spla_Scalar_get_uint(self.hnd, C.byref(v))
        else:
            v = C.c_float()
            lib.spla_Scalar_get_float(self.hnd, C.byref(v))
        return v.value


class Descriptor(Object):
    PUSH, PULL, PUSH_PULL = 0, 1, 2

    def __init__(self, early_exit=False, struct_only=False, traversal_mode=None, front_factor=None, deterministic=False, reference_order=False):
        h = _vp()
        _check(_c().spla_Descriptor_make(C.byref(h)), "spla_Descriptor_make")
        super().__init__(h)
        _c().spla_Descriptor_set_early_exit(h, 1 if early_exit else 0)
        _c().spla_Descriptor_set_struct_only(h, 1 if struct_only else 0)
        if deterministic:  # fp32 sums on the fixed-order kernels: two runs of the call are bit-identical
            _check(_c().spla_Descriptor_set_deterministic(h, 1), "spla_Descriptor_set_deterministic")
        if reference_order:  # fp32 sums folded as the reference CPU backend folds them: bit-identical to it
            _check(_c().spla_Descriptor_set_reference_order(h, 1), "spla_Descriptor_set_reference_order")
        if traversal_mode is not None:
            _check(_c().spla_Descriptor_set_traversal_mode(h, int(traversal_mode)), "spla_Descriptor_set_traversal_mode")
        if front_factor is not None:
            _check(_c().spla_Descriptor_set_front_factor(h, float(front_factor)), "spla_Descriptor_set_front_factor")


def _desc(desc):
    return desc.hnd if desc is not None else None


class Vector(Object):
    def __init__(self, shape, dtype=INT):
        h = _vp()
        _check(_c().spla_Vector_make(C.byref(h), int(shape), dtype._hnd()), "spla_Vector_make")
        if not h:
            raise SplaError(6, "spla_Vector_make returned a null handle")
        super().__init__(h)
        self.dtype, self.n_rows = dtype, int(shape)

    @property
    def shape(self):
        return (self.n_rows, 1)

    def set_format(self, fmt):
        _check(_c().spla_Vector_set_format(self.hnd, int(fmt)), "spla_Vector_set_format")

    def set_fill_value(self, value):
        s = Scalar(self.dtype, value)
        _check(_c().spla_Vector_set_fill_value(self.hnd, s.hnd), "spla_Vector_set_fill_value")

    def set(self, i, v):
        lib = _c()
        if self.dtype is INT:
            st = lib.spla_Vector_set_int(self.hnd, i, int(np.int32(v)))
        elif self.dtype is UINT:
            st = lib.spla_Vector_set_uint(self.hnd, i, int(np.uint32(v)))
        else:
            st = lib.spla_Vector_set_float(self.hnd, i, float(v))
        _check(st, "spla_Vector_set")

    def get(self, i):
        lib = _c()
        if self.dtype is INT:
            v = C.c_int()
            st = lib.spla_Vector_get_int(self.hnd, i, C.byref(v))
        elif self.dtype is UINT:
            v = C.c_uint()
            st = lib.spla_Vector_get_uint(self.hnd, i, C.byref(v))
        else:
            v = C.c_float()
            st = lib.spla_Vector_get_float(self.hnd, i, C.byref(v))
        _check(st, "spla_Vector_get")
        return v.value

    def build(self, I, V):
        I = np.ascontiguousarray(I, dtype=np.uint32)
        V = np.ascontiguousarray(V, dtype=self.dtype._np)
        vi, vv = MemView(I), MemView(V)
        _check(_c().spla_Vector_build(self.hnd, vi.hnd, vv.hnd), "spla_Vector_build")
        return self

    def read(self):
        k, x = _vp(), _vp()
        _check(_c().spla_Vector_read(self.hnd, C.byref(k), C.byref(x)), "spla_Vector_read")
        mk, mx = MemView(hnd=k), MemView(hnd=x)
        return mk.to_numpy(np.uint32), mx.to_numpy(self.dtype._np)

    def clear(self):
        _check(_c().spla_Vector_clear(self.hnd), "spla_Vector_clear")

    def to_lists(self):
        I, V = self.read()
        return I.tolist(), V.tolist()

    def to_dense(self):
        """All n_rows values, absent entries = fill value (element-wise get, as pyspla's to_list)."""
        return np.array([self.get(i) for i in range(self.n_rows)], dtype=self.dtype._np)

    @classmethod
    def from_lists(cls, I, V, shape, dtype=INT):
        return cls(shape, dtype).build(I, V)

    @classmethod
    def dense(cls, shape, dtype=INT, fill_value=0):
        v = cls(shape, dtype)
        for i in range(shape):
            v.set(i, fill_value)
        return v

    def assign(self, mask, value, op_assign, op_select, desc=None):
        """self[i] = op_assign(self[i], value) where op_select(mask[i]) (reference python/pyspla/vector.py:640-683)."""
        if not isinstance(value, Scalar):
            value = Scalar(self.dtype, value)
        _check(_c().spla_Exec_v_assign_masked(self.hnd, mask.hnd, value.hnd, op_assign.hnd, op_select.hnd, _desc(desc), None), "spla_Exec_v_assign_masked")
        return self

    def eadd_fdb(self, v, op, fdb=None, desc=None):
        """self[i] = op(self[i], v[i]); returns the feedback vector holding the entries that changed (reference exec_v_eadd_fdb)."""
        if fdb is None:
            fdb = Vector(self.n_rows, self.dtype)
        _check(_c().spla_Exec_v_eadd_fdb(self.hnd, v.hnd, fdb.hnd, op.hnd, _desc(desc), None), "spla_Exec_v_eadd_fdb")
        return fdb

    def count_mf(self, desc=None) -> int:
        """number of stored (sparse formats) / non-fill (dense formats) entries (reference exec_v_count_mf)."""
        out = Scalar(UINT, 0)
        _check(_c().spla_Exec_v_count_mf(out.hnd, self.hnd, _desc(desc), None), "spla_Exec_v_count_mf")
        return int(out.get())

    def eadd(self, v, op, out=None, desc=None):
        """out = op(self, v) element-wise (reference python/pyspla/vector.py eadd / exec_v_eadd)."""
        if out is None:
            out = Vector(self.n_rows, self.dtype)
        _check(_c().spla_Exec_v_eadd(out.hnd, self.hnd, v.hnd, op.hnd, _desc(desc), None), "spla_Exec_v_eadd")
        return out

    def reduce(self, op, init=None, desc=None):
        """fold of the stored values from init (reference python/pyspla/vector.py reduce / exec_v_reduce)."""
        out = Scalar(self.dtype, 0)
        if init is None:
            init = Scalar(self.dtype, 0)
        elif not isinstance(init, Scalar):
            init = Scalar(self.dtype, init)
        _check(_c().spla_Exec_v_reduce(out.hnd, init.hnd, self.hnd, op.hnd, _desc(desc), None), "spla_Exec_v_reduce")
        return out.get()

    def map(self, op, out=None, desc=None):
        """out = op(self) element-wise: stored entries of a sparse vector, every index of a dense one (reference python/pyspla/vector.py:690-720)."""
        if out is None:
            out = Vector(self.n_rows, self.dtype)
        _check(_c().spla_Exec_v_map(out.hnd, self.hnd, op.hnd, _desc(desc), None), "spla_Exec_v_map")
        return out

    def vxm(self, mask, M, op_mult, op_add, op_select, out=None, init=None, desc=None):
        """r = self^T x M over (op_mult, op_add), columns selected by mask (reference python/pyspla/vector.py:476-546)."""
        if out is None:
            out = Vector(M.n_cols, self.dtype)
        if init is None:
            init = Scalar(self.dtype, 0)
        _check(_c().spla_Exec_vxm_masked(out.hnd, mask.hnd, self.hnd, M.hnd, op_mult.hnd, op_add.hnd, op_select.hnd, init.hnd,
                                         _desc(desc), None), "spla_Exec_vxm_masked")
        return out


class Matrix(Object):
    def __init__(self, shape, dtype=INT):
        h = _vp()
        _check(_c().spla_Matrix_make(C.byref(h), int(shape[0]), int(shape[1]), dtype._hnd()), "spla_Matrix_make")
        if not h:
            raise SplaError(6, "spla_Matrix_make returned a null handle")
        super().__init__(h)
        self.dtype, self.n_rows, self.n_cols = dtype, int(shape[0]), int(shape[1])

    @property
    def shape(self):
        return (self.n_rows, self.n_cols)

    def set_format(self, fmt):
        _check(_c().spla_Matrix_set_format(self.hnd, int(fmt)), "spla_Matrix_set_format")

    def set(self, i, j, v):
        lib = _c()
        if self.dtype is INT:
            st = lib.spla_Matrix_set_int(self.hnd, i, j, int(np.int32(v)))
        elif self.dtype is UINT:
            st = lib.spla_Matrix_set_uint(self.hnd, i, j, int(np.uint32(v)))
        else:
            st = lib.spla_Matrix_set_float(self.hnd, i, j, float(v))
        _check(st, "spla_Matrix_set")

    def get(self, i, j):
        lib = _c()
        if self.dtype is INT:
            v = C.c_int()
            st = lib.spla_Matrix_get_int(self.hnd, i, j, C.byref(v))
        elif self.dtype is UINT:
            v = C.c_uint()
            st = lib.spla_Matrix_get_uint(self.hnd, i, j, C.byref(v))
        else:
            v = C.c_float()
            st = lib.spla_Matrix_get_float(self.hnd, i, j, C.byref(v))
        _check(st, "spla_Matrix_get")
        return v.value

    def build(self, I, J, V):
        I = np.ascontiguousarray(I, dtype=np.uint32)
        J = np.ascontiguousarray(J, dtype=np.uint32)
        V = np.ascontiguousarray(V, dtype=self.dtype._np)
        vi, vj, vv = MemView(I), MemView(J), MemView(V)
        _check(_c().spla_Matrix_build(self.hnd, vi.hnd, vj.hnd, vv.hnd), "spla_Matrix_build")
        return self

    def read(self):
        a, b, c = _vp(), _vp(), _vp()
        _check(_c().spla_Matrix_read(self.hnd, C.byref(a), C.byref(b), C.byref(c)), "spla_Matrix_read")
        ma, mb, mc = MemView(hnd=a), MemView(hnd=b), MemView(hnd=c)
        return ma.to_numpy(np.uint32), mb.to_numpy(np.uint32), mc.to_numpy(self.dtype._np)

    def clear(self):
        _check(_c().spla_Matrix_clear(self.hnd), "spla_Matrix_clear")

    def to_lists(self):
        I, J, V = self.read()
        return I.tolist(), J.tolist(), V.tolist()

    @classmethod
    def from_lists(cls, I, J, V, shape, dtype=INT):
        return cls(shape, dtype).build(I, J, V)

    @classmethod
    def from_csr(cls, Ap, Aj, Ax, shape, dtype=INT):
        Ap = np.asarray(Ap).astype(np.int64)
        I = np.repeat(np.arange(shape[0], dtype=np.uint32), np.diff(Ap))
        return cls(shape, dtype).build(I, Aj, Ax)

    def mxv(self, mask, v, op_mult, op_add, op_select, out=None, init=None, desc=None):
        """r = self x v over (op_mult, op_add), rows selected by mask (reference python/pyspla/matrix.py:939-1015)."""
        if out is None:
            out = Vector(self.n_rows, self.dtype)
        if init is None:
            init = Scalar(self.dtype, 0)
        _check(_c().spla_Exec_mxv_masked(out.hnd, mask.hnd, self.hnd, v.hnd, op_mult.hnd, op_add.hnd, op_select.hnd, init.hnd,
                                         _desc(desc), None), "spla_Exec_mxv_masked")
        return out

    def mxm(self, M, op_mult, op_add, out=None, init=None, desc=None):
        """R = self x M (reference python/pyspla/matrix.py:574-653)."""
        if out is None:
            out = Matrix((self.n_rows, M.n_cols), self.dtype)
        if init is None:
            init = Scalar(self.dtype, 0)
        _check(_c().spla_Exec_mxm(out.hnd, self.hnd, M.hnd, op_mult.hnd, op_add.hnd, init.hnd, _desc(desc), None), "spla_Exec_mxm")
        return out

    def mxmT(self, mask, M, op_mult, op_add, op_select, out=None, init=None, desc=None):
        """R = (self x M^T) . mask (reference python/pyspla/matrix.py:655-764)."""
        if out is None:
            out = Matrix((self.n_rows, M.n_rows), self.dtype)
        if init is None:
            init = Scalar(self.dtype, 0)
        _check(_c().spla_Exec_mxmT_masked(out.hnd, mask.hnd, self.hnd, M.hnd, op_mult.hnd, op_add.hnd, op_select.hnd, init.hnd,
                                          _desc(desc), None), "spla_Exec_mxmT_masked")
        return out

    def reduce(self, op, init=None, desc=None):
        """fold of the stored values from init (reference python/pyspla/matrix.py reduce / exec_m_reduce)."""
        out = Scalar(self.dtype, 0)
        if init is None:
            init = Scalar(self.dtype, 0)
        elif not isinstance(init, Scalar):
            init = Scalar(self.dtype, init)
        _check(_c().spla_Exec_m_reduce(out.hnd, init.hnd, self.hnd, op.hnd, _desc(desc), None), "spla_Exec_m_reduce")
        return out.get()

    def transpose(self, op_apply=None, out=None, desc=None):
        """out = op_apply(self)^T (reference python/pyspla/matrix.py:1300-1338 / exec_m_transpose); op_apply defaults to IDENTITY."""
        if out is None:
            out = Matrix((self.n_cols, self.n_rows), self.dtype)
        if op_apply is None:
            op_apply = self.dtype.IDENTITY
        _check(_c().spla_Exec_m_transpose(out.hnd, self.hnd, op_apply.hnd, _desc(desc), None), "spla_Exec_m_transpose")
        return out


# ---- graph algorithms (reference C API include/spla.h:360-363) ----
def bfs(A: Matrix, s: int, desc=None) -> Vector:
    """levels from 1 at the source, 0 = unreached; A is the INT adjacency matrix, edges i -> j at A[i, j]."""
    v = Vector(A.n_rows, INT)
    _check(_c().spla_Algorithm_bfs(v.hnd, A.hnd, int(s), _desc(desc)), "spla_Algorithm_bfs")
    return v


def sssp(A: Matrix, s: int, desc=None) -> Vector:
    """shortest path lengths over FLOAT weights; unreachable = FLT_MAX (the result's fill value)."""
    v = Vector(A.n_rows, FLOAT)
    _check(_c().spla_Algorithm_sssp(v.hnd, A.hnd, int(s), _desc(desc)), "spla_Algorithm_sssp")
    return v


def pr(A: Matrix, alpha=0.85, eps=1e-6, desc=None) -> Vector:
    """PageRank by power iteration; A must already hold alpha / outdeg(j) at [i, j] for every edge j -> i."""
    p = Vector(A.n_rows, FLOAT)
    h = _vp(p.hnd.value if isinstance(p.hnd, _vp) else p.hnd)
    _check(_c().spla_Algorithm_pr(C.byref(h), A.hnd, float(alpha), float(eps), _desc(desc)), "spla_Algorithm_pr")
    p.hnd = h  # the C call consumed the old handle's reference and handed one to the result
    return p


def tc(L: Matrix, desc=None):
    """triangle count of the graph whose strict lower triangle is L (INT); returns (count, B = (L x L^T) .* L)."""
    B = Matrix((L.n_rows, L.n_cols), INT)
    n = C.c_int(0)
    _check(_c().spla_Algorithm_tc(C.byref(n), L.hnd, B.hnd, _desc(desc)), "spla_Algorithm_tc")
    return int(n.value), B


def load_mtx(path, offset_indices=True, make_undirected=True, remove_loops=True):
    """MatrixMarket coordinate file -> (n_rows, n_cols, Ai, Aj): sorted, de-duplicated edge list (reference MtxLoader::load semantics)."""
    h = _vp()
    _check(_c().spla_MtxLoader_load(C.byref(h), str(path).encode(), int(offset_indices), int(make_undirected), int(remove_loops)), "spla_MtxLoader_load")
    try:
        nr, nc, nv = C.c_uint(), C.c_uint(), C.c_size_t()
        pi, pj = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        _check(_c().spla_MtxLoader_get(h, C.byref(nr), C.byref(nc), C.byref(nv), C.byref(pi), C.byref(pj)), "spla_MtxLoader_get")
        n = nv.value
        Ai = np.ctypeslib.as_array(pi, shape=(n,)).copy() if n else np.empty(0, np.uint32)
        Aj = np.ctypeslib.as_array(pj, shape=(n,)).copy() if n else np.empty(0, np.uint32)
        return nr.value, nc.value, Ai, Aj
    finally:
        _c().spla_MtxLoader_delete(h)
