"""The vector operations between two hot-path calls of a traversal: v_assign_masked, v_eadd_fdb,
v_count_mf.  CPU: oracle == genuine reference; GPU: HIP (kernel-level ABI and operator API) == oracle, bit-exact."""
import numpy as np
import pytest

from helpers import NP, random_mask, random_values

CASES = [("SECOND", "NQZERO"), ("PLUS", "GTZERO"), ("MIN", "EQZERO"), ("MAX", "ALWAYS"), ("MULT", "LEZERO"), ("FIRST", "NEVER")]


def _inputs(rng, t, n=777):
    r = random_values(rng, n, t, zeros=0.3)
    mask = random_mask(rng, n, t)
    idx = np.sort(rng.permutation(n)[: n // 3]).astype(np.uint32)
    v = random_values(rng, n, t, zeros=0.3)
    return r, mask, idx, v


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_oracle_equals_reference(oracle, reference, t):
    rng = np.random.default_rng(71)
    for op, sel in CASES:
        r, mask, idx, v = _inputs(rng, t)
        assert _same(oracle.v_assign_masked(t, r, mask, 5, op, sel), reference.v_assign_masked(t, r, mask, 5, op, sel))
        assert _same(oracle.v_assign_masked(t, r, mask[idx], 5, op, sel, mask_idx=idx), reference.v_assign_masked(t, r, mask[idx], 5, op, sel, mask_idx=idx))
        a, b = oracle.v_eadd_fdb(t, r, v, op, fdb_fill=9), reference.v_eadd_fdb(t, r, v, op, fdb_fill=9)
        assert _same(a[0], b[0]) and _same(a[1], b[1])
        a, b = oracle.v_eadd_fdb(t, r, v[idx], op, v_idx=idx), reference.v_eadd_fdb(t, r, v[idx], op, v_idx=idx)
        assert _same(a[0], b[0]) and np.array_equal(a[1][0], b[1][0]) and _same(a[1][1], b[1][1])
        assert oracle.v_count_mf(t, r, 0) == reference.v_count_mf(t, r, 0)
        assert oracle.v_count_mf(t, r[idx], idx=idx, n=len(r)) == reference.v_count_mf(t, r[idx], idx=idx, n=len(r)) == len(idx)


@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_hip_kernels_equal_oracle(oracle, t):
    import ctypes as C
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import binop_id, scalar_bits, selop_id, type_id
    rng = np.random.default_rng(73)
    tid = type_id(t)
    for op, sel in CASES:
        r, mask, idx, v = _inputs(rng, t, n=5000)
        n = len(r)
        # operands stay referenced until the end of the iteration: a temporary's block would be handed to the next dev()
        d_mask, d_idx, d_midx, d_v, d_vidx, d_r0 = dev(mask), dev(idx), dev(mask[idx]), dev(v), dev(v[idx]), dev(r)
        # assign, dense and sparse mask
        dr = dev(r.copy())
        _lib.call("splahip_v_assign_masked_dense", tid, n, _ptr(dr), _ptr(d_mask), scalar_bits(5, tid), binop_id(op), selop_id(sel), current_stream_ptr())
        assert _same(host(dr, NP[t]), oracle.v_assign_masked(t, r, mask, 5, op, sel))
        dr = dev(r.copy())
        _lib.call("splahip_v_assign_masked_sparse", tid, len(idx), _ptr(dr), _ptr(d_idx), _ptr(d_midx), scalar_bits(5, tid), binop_id(op), selop_id(sel), current_stream_ptr())
        assert _same(host(dr, NP[t]), oracle.v_assign_masked(t, r, mask[idx], 5, op, sel, mask_idx=idx))
        # eadd_fdb dense
        dr, dfdb = dev(r.copy()), torch.empty(n, dtype=d_r0.dtype, device="cuda")
        _lib.call("splahip_v_eadd_fdb_dense", tid, n, _ptr(dr), _ptr(d_v), _ptr(dfdb), scalar_bits(9, tid), binop_id(op), current_stream_ptr())
        want = oracle.v_eadd_fdb(t, r, v, op, fdb_fill=9)
        assert _same(host(dr, NP[t]), want[0]) and _same(host(dfdb, NP[t]), want[1])
        # eadd_fdb sparse: feedback in the operand's stored order
        dr = dev(r.copy())
        fi, fx = torch.empty(len(idx), dtype=torch.int32, device="cuda"), torch.empty(len(idx), dtype=d_r0.dtype, device="cuda")
        cnt = C.c_uint32()
        _lib.call("splahip_v_eadd_fdb_sparse", tid, len(idx), _ptr(dr), _ptr(d_idx), _ptr(d_vidx), _ptr(fi), _ptr(fx), C.byref(cnt), binop_id(op), current_stream_ptr())
        want = oracle.v_eadd_fdb(t, r, v[idx], op, v_idx=idx)
        assert _same(host(dr, NP[t]), want[0]) and cnt.value == len(want[1][0])
        assert np.array_equal(host(fi, np.uint32)[:cnt.value], want[1][0]) and _same(host(fx, NP[t])[:cnt.value], want[1][1])
        # count
        cnt = C.c_uint32()
        _lib.call("splahip_v_count_mf_dense", tid, n, _ptr(d_r0), scalar_bits(0, tid), C.byref(cnt), current_stream_ptr())
        assert cnt.value == oracle.v_count_mf(t, r, 0)


@pytest.mark.gpu
def test_bfs_iteration_through_the_api(oracle):
    """One pull-BFS iteration written like the reference's bfs() (src/algorithm.cpp:88-103): assign depth, mxv with early exit,
    count the new frontier — every step on the device through spla_Exec_*; checked against a plain numpy BFS."""
    from spla_amd import api
    rng = np.random.default_rng(5)
    n = 600
    dense = rng.random((n, n)) < 0.01
    np.fill_diagonal(dense, False)
    I, J = np.nonzero(dense)
    A = api.Matrix.from_lists(I, J, np.ones(len(I), np.int32), (n, n), api.INT)
    depth = api.Vector.dense(n, api.INT, 0)
    frontier = api.Vector.from_lists([3], [1], n, api.INT)
    level, ref_depth, ref_front = 1, np.zeros(n, np.int32), np.zeros(n, bool)
    ref_front[3] = True
    desc = api.Descriptor(early_exit=True)
    while True:
        depth.assign(frontier, level, api.INT.SECOND, api.INT.NQZERO)  # depth[frontier] = level
        ref_depth[ref_front] = level
        new = A.mxv(depth, frontier, api.INT.BAND, api.INT.BOR, api.INT.EQZERO, init=api.Scalar(api.INT, 0), desc=desc)
        ref_new = (dense[:, ref_front].any(axis=1)) & (ref_depth == 0)
        size = new.count_mf()
        assert size == int(ref_new.sum())
        if size == 0:
            break
        frontier, ref_front, level = new, ref_new, level + 1
    assert np.array_equal(depth.to_dense(), ref_depth)


@pytest.mark.gpu
def test_eadd_fdb_api_dense_and_sparse(oracle):
    from spla_amd import api
    rng = np.random.default_rng(8)
    n = 900
    r_np = rng.integers(0, 50, n).astype(np.float32)
    v_np = rng.integers(0, 50, n).astype(np.float32)
    # dense x dense
    r = api.Vector.from_lists(np.arange(n), r_np, n, api.FLOAT)
    r.set_format(1)
    v = api.Vector.from_lists(np.arange(n), v_np, n, api.FLOAT)
    v.set_format(1)
    fdb = api.Vector(n, api.FLOAT)
    fdb.set_fill_value(1e9)
    r.eadd_fdb(v, api.FLOAT.MIN, fdb=fdb)
    want_r, want_f = oracle.v_eadd_fdb("FLOAT", r_np, v_np, "MIN", fdb_fill=1e9)
    assert np.array_equal(r.to_dense(), want_r)
    I, V = fdb.read()
    assert np.array_equal(I, np.nonzero(want_f != np.float32(1e9))[0]) and np.array_equal(V, want_f[want_f != np.float32(1e9)])
    # dense r, sparse v
    idx = np.sort(rng.permutation(n)[:200]).astype(np.uint32)
    r = api.Vector.from_lists(np.arange(n), r_np, n, api.FLOAT)
    r.set_format(1)
    v = api.Vector.from_lists(idx, v_np[idx], n, api.FLOAT)
    fdb = r.eadd_fdb(v, api.FLOAT.MIN)
    want_r, (wi, wx) = oracle.v_eadd_fdb("FLOAT", r_np, v_np[idx], "MIN", v_idx=idx)
    assert np.array_equal(r.to_dense(), want_r)
    I, V = fdb.read()
    assert np.array_equal(I, wi) and np.array_equal(V, wx)
    assert fdb.count_mf() == len(wi)
