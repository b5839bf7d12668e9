"""PageRank-step / triangle-count-tail operations (v_eadd, v_reduce, m_reduce) and the four graph algorithms over exec_*
(SURVEY.md 8f ranks 2-3).  CPU: oracle == genuine reference (operations bit for bit; algorithms through the reference's own
spla_Algorithm_*).  GPU: HIP == oracle through the kernel-level ABI and the operator API."""
import numpy as np
import pytest

from helpers import NP

FLT_MAX = np.float32(np.finfo(np.float32).max)


def vals(rng, n, t, lo=1, hi=6, zeros=0.0):
    """small integer-valued entries of type t (sums and short products stay exact in fp32 and never overflow int32)"""
    x = rng.integers(lo, hi, n).astype(NP[t])
    if zeros > 0:
        x[rng.random(n) < zeros] = 0
    return x


def reduce_len(op, n):
    """how many elements a fold with op may take before the running value leaves the exactly representable range"""
    return {"MULT": 12, "MINUS_POW2": 4}.get(op, n)


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def sym_graph(rng, n, deg, weights=False):
    """symmetric pattern without self-loops as CSR (rows sorted by column)"""
    m = n * deg // 2
    a, b = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = a != b
    a, b = a[keep], b[keep]
    key = np.unique(np.concatenate([a * n + b, b * n + a]))
    I, J = (key // n).astype(np.uint32), (key % n).astype(np.uint32)
    Ap = np.zeros(n + 1, np.uint32)
    np.cumsum(np.bincount(I, minlength=n), out=Ap[1:])
    if not weights:
        return Ap, J
    w = {}
    lo, hi = np.minimum(I, J).astype(np.int64), np.maximum(I, J).astype(np.int64)
    base = rng.integers(1, 64, n * 4).astype(np.float32) / np.float32(4)
    X = base[(lo * 31 + hi * 17) % len(base)]  # symmetric weights
    return Ap, J, X.astype(np.float32)


def lower_triangle(Ap, Aj):
    n = len(Ap) - 1
    rows = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap.astype(np.int64)))
    keep = Aj < rows
    Lp = np.zeros(n + 1, np.uint32)
    np.cumsum(np.bincount(rows[keep], minlength=n), out=Lp[1:])
    return Lp, Aj[keep]


def pagerank_matrix(Ap, Aj, alpha=0.85):
    """A[i, j] = alpha / deg(j) on a symmetric pattern (reference examples/pr.cpp:81-88)"""
    deg = np.diff(Ap.astype(np.int64))
    return (np.float32(alpha) / deg[Aj].astype(np.float32)).astype(np.float32)


OPS = ["PLUS", "MIN", "MAX", "MULT", "MINUS", "MINUS_POW2", "FIRST", "SECOND"]


@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_oracle_ops_equal_reference(oracle, reference, t):
    rng = np.random.default_rng(17)
    n = 400
    for op in OPS:
        u, v = vals(rng, n, t, zeros=0.2), vals(rng, n, t, zeros=0.2)
        assert _same(oracle.v_eadd(t, u, v, op), reference.v_eadd(t, u, v, op))
        ui = np.sort(rng.permutation(n)[:120]).astype(np.uint32)
        vi = np.sort(rng.permutation(n)[:150]).astype(np.uint32)
        a = oracle.v_eadd(t, u[ui], v[vi], op, u_idx=ui, v_idx=vi, u_fill=2, v_fill=3)
        b = reference.v_eadd(t, u[ui], v[vi], op, u_idx=ui, v_idx=vi, u_fill=2, v_fill=3)
        assert np.array_equal(a[0], b[0]) and _same(a[1], b[1])
        small = vals(rng, 40, t, 1, 3)[:reduce_len(op, 40)]
        assert _same(oracle.reduce(t, small, op, 2), reference.reduce(t, small, op, 2))
        k = min(9, len(small))
        assert _same(oracle.reduce(t, small[:k], op, 2), reference.reduce(t, small[:k], op, 2, idx=np.arange(k, dtype=np.uint32) * 3, n=40))


def test_oracle_m_reduce_equals_reference(oracle, reference):
    rng = np.random.default_rng(19)
    Ap, Aj = sym_graph(rng, 60, 6)
    for t in ("INT", "FLOAT"):
        Ax = vals(rng, len(Aj), t, 1, 5)
        for op in ("PLUS", "MAX", "MIN"):
            assert _same(oracle.reduce(t, Ax, op, 1), reference.m_reduce(t, 60, 60, Ap, Aj, Ax, op, 1))


def test_oracle_algorithms_equal_reference(oracle, reference):
    rng = np.random.default_rng(23)
    n = 300
    Ap, Aj, Ax = sym_graph(rng, n, 5, weights=True)
    src = int(np.argmax(np.diff(Ap.astype(np.int64))))
    want = oracle.bfs(n, Ap, Aj, src)
    for mode in ("push", "pull", "push_pull"):
        assert np.array_equal(reference.bfs(n, Ap, Aj, src, mode=mode), want), mode
    want = oracle.sssp(n, Ap, Aj, Ax, src)
    for mode in ("push", "pull", "push_pull"):
        assert _same(reference.sssp(n, Ap, Aj, Ax, src, mode=mode), want), mode
    Px = pagerank_matrix(Ap, Aj)
    assert _same(reference.pr(n, Ap, Aj, Px, 0.85, 1e-5), oracle.pr(n, Ap, Aj, Px, 0.85, 1e-5))
    Lp, Lj = lower_triangle(Ap, Aj)
    got, want = reference.tc(n, Lp, Lj), oracle.tc(n, Lp, Lj)
    assert got[0] == want[0] and all(np.array_equal(a, b) for a, b in zip(got[1], want[1]))
    # cross-check against an independent count: trace(A^3) / 6 on the dense pattern
    D = np.zeros((n, n), np.int64)
    D[np.repeat(np.arange(n), np.diff(Ap.astype(np.int64))), Aj] = 1
    assert want[0] == int(np.trace(D @ D @ D)) // 6


def test_directed_bfs_push_and_pull_disagree_like_the_reference(oracle, reference):
    """push follows A[i, j] as i -> j, pull reads it as j -> i (src/algorithm.cpp:98-101): on a directed chain they differ"""
    n = 5
    Ap, Aj = np.array([0, 1, 2, 3, 4, 4], np.uint32), np.array([1, 2, 3, 4], np.uint32)  # 0->1->2->3->4
    assert np.array_equal(reference.bfs(n, Ap, Aj, 0, mode="push"), oracle.bfs(n, Ap, Aj, 0))
    assert np.array_equal(reference.bfs(n, Ap, Aj, 4, mode="pull"), oracle.bfs(n, Ap, Aj, 4, pull=True))
    assert np.array_equal(oracle.bfs(n, Ap, Aj, 0), [1, 2, 3, 4, 5]) and np.array_equal(oracle.bfs(n, Ap, Aj, 4, pull=True), [5, 4, 3, 2, 1])


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("t", ["INT", "UINT", "FLOAT"])
def test_hip_eadd_and_reduce_equal_oracle(oracle, t):
    import ctypes as C
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import binop_id, scalar_bits, type_id
    rng = np.random.default_rng(29)
    tid = type_id(t)
    n = 70001
    for op in OPS:
        u, v = vals(rng, n, t, zeros=0.2), vals(rng, n, t, zeros=0.2)
        du, dv = dev(u), dev(v)
        dr = torch.empty_like(du)
        _lib.call("splahip_v_eadd_dense", tid, n, _ptr(dr), _ptr(du), _ptr(dv), binop_id(op), current_stream_ptr())
        assert _same(host(dr, NP[t]), oracle.v_eadd(t, u, v, op))
        ui = np.sort(rng.permutation(n)[:20000]).astype(np.uint32)
        vi = np.sort(rng.permutation(n)[:31000]).astype(np.uint32)
        dui, dux, dvi, dvx = dev(ui), dev(u[ui]), dev(vi), dev(v[vi])
        ri = torch.empty(len(ui) + len(vi), dtype=torch.int32, device="cuda")
        rx = torch.empty(len(ui) + len(vi), dtype=du.dtype, device="cuda")
        cnt = C.c_uint32()
        _lib.call("splahip_v_eadd_sparse", tid, len(ui), _ptr(dui), _ptr(dux), scalar_bits(2, tid), len(vi), _ptr(dvi), _ptr(dvx), scalar_bits(3, tid),
                  _ptr(ri), _ptr(rx), C.byref(cnt), binop_id(op), current_stream_ptr())
        wi, wx = oracle.v_eadd(t, u[ui], v[vi], op, u_idx=ui, v_idx=vi, u_fill=2, v_fill=3)
        assert cnt.value == len(wi) and np.array_equal(host(ri, np.uint32)[:cnt.value], wi) and _same(host(rx, NP[t])[:cnt.value], wx)
        # reduce: small positive values; MULT on a short prefix so nothing overflows / loses exactness
        x = vals(rng, n, t, 1, 3)
        m = reduce_len(op, n)  # (fp32 PLUS of n small integers stays below 2^24: exact in any order)
        dx = dev(x)
        out = C.c_uint32()
        _lib.call("splahip_reduce", tid, m, _ptr(dx), scalar_bits(2, tid), C.byref(out), binop_id(op), current_stream_ptr())
        assert _same(np.array([out.value], np.uint32), np.array([oracle.reduce(t, x[:m], op, 2)]).view(np.uint32)), (op, m)


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["MIN", "MAX"])
def test_hip_reduce_fp32_min_max_with_nan(oracle, op):
    """the reduce tree under fp32 MIN / MAX: NaN elements are never picked by the reference's fold from `s`, a NaN `s` sticks (src/op.cpp:151-155)"""
    import ctypes as C
    from gpu_utils import dev
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import binop_id, type_id
    rng = np.random.default_rng(37)
    x = (rng.integers(-4000, 4000, 300001) / 8.0).astype(np.float32)
    x[rng.random(len(x)) < 0.1] = np.nan
    x[:9] = np.nan  # whole quads of NaN at the start of the first lane's share
    dx = dev(x)
    for s in (np.float32(0.5), np.float32(np.inf), np.float32(-np.inf), np.float32(np.nan)):
        out = C.c_uint32()
        _lib.call("splahip_reduce", type_id("FLOAT"), len(x), _ptr(dx), int(np.array([s]).view(np.uint32)[0]), C.byref(out), binop_id(op), current_stream_ptr())
        want = np.array([oracle.reduce("FLOAT", x, op, s)], np.float32).view(np.uint32)[0]
        assert out.value == want, (op, s, hex(out.value), hex(want))


@pytest.mark.gpu
def test_hip_reduce_fp32_plus_within_tolerance(oracle):
    """fp32 PLUS through a fixed tree instead of the sequential fold: <= 1e-5 relative on positive data (tolerance of the north star)"""
    import ctypes as C
    from gpu_utils import dev
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import binop_id, scalar_bits, type_id
    rng = np.random.default_rng(31)
    x = rng.random(1 << 20, dtype=np.float32)
    dx = dev(x)
    out = C.c_uint32()
    _lib.call("splahip_reduce", type_id("FLOAT"), len(x), _ptr(dx), scalar_bits(0.5, 2), C.byref(out), binop_id("PLUS"), current_stream_ptr())
    got = np.array([out.value], np.uint32).view(np.float32)[0]
    exact = float(np.sum(x.astype(np.float64)) + 0.5)
    assert abs(got - exact) <= 1e-5 * exact
    # the tree is at least as close to the exact sum as the reference's sequential fp32 fold
    assert abs(got - exact) <= abs(float(oracle.reduce("FLOAT", x, "PLUS", 0.5)) - exact) + 1e-6 * exact


@pytest.mark.gpu
def test_algorithms_on_the_device_equal_oracle(oracle):
    from spla_amd import api
    rng = np.random.default_rng(37)
    n = 3000
    Ap, Aj, Ax = sym_graph(rng, n, 6, weights=True)
    I = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap.astype(np.int64)))
    src = int(np.argmax(np.diff(Ap.astype(np.int64))))

    A = api.Matrix.from_lists(I, Aj, np.ones(len(Aj), np.int32), (n, n), api.INT)
    want = oracle.bfs(n, Ap, Aj, src)
    for mode in (api.Descriptor.PUSH, api.Descriptor.PULL, api.Descriptor.PUSH_PULL):
        got = api.bfs(A, src, api.Descriptor(traversal_mode=mode, front_factor=0.05))
        assert np.array_equal(got.to_dense(), want), mode
    assert np.array_equal(api.bfs(A, src).to_dense(), want)  # NULL descriptor = defaults

    W = api.Matrix.from_lists(I, Aj, Ax, (n, n), api.FLOAT)
    want = oracle.sssp(n, Ap, Aj, Ax, src)
    for mode in (api.Descriptor.PUSH, api.Descriptor.PULL, api.Descriptor.PUSH_PULL):
        got = api.sssp(W, src, api.Descriptor(traversal_mode=mode, front_factor=0.05)).to_dense()
        assert _same(got, want), mode  # the fp32 Bellman-Ford fixed point does not depend on the schedule

    Px = pagerank_matrix(Ap, Aj)
    P = api.Matrix.from_lists(I, Aj, Px, (n, n), api.FLOAT)
    got = api.pr(P, 0.85, 1e-6).to_dense()
    want = oracle.pr(n, Ap, Aj, Px, 0.85, 1e-6)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-7)  # both stop within eps of the fixed point; folds differ by rounding

    Lp, Lj = lower_triangle(Ap, Aj)
    LI = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Lp.astype(np.int64)))
    L = api.Matrix.from_lists(LI, Lj, np.ones(len(Lj), np.int32), (n, n), api.INT)
    count, B = api.tc(L)
    want_count, (Bp, Bj, Bx) = oracle.tc(n, Lp, Lj)
    assert count == want_count
    assert B.reduce(api.INT.PLUS) == want_count and B.reduce(api.INT.MAX, init=0) == int(Bx.max(initial=0))


@pytest.mark.gpu
def test_eadd_and_reduce_through_the_api(oracle):
    from spla_amd import api
    rng = np.random.default_rng(41)
    n = 1000
    u_np, v_np = rng.integers(1, 9, n).astype(np.float32), rng.integers(1, 9, n).astype(np.float32)
    ui = np.sort(rng.permutation(n)[:300]).astype(np.uint32)
    vi = np.sort(rng.permutation(n)[:200]).astype(np.uint32)
    # sparse x sparse (both built from lists: COO is the valid format), fill values take part
    u, v = api.Vector(n, api.FLOAT), api.Vector(n, api.FLOAT)
    u.set_fill_value(2.0)  # before build: set_fill_value drops the content (reference src/core/tvector.hpp:131-140)
    v.set_fill_value(3.0)
    u.build(ui, u_np[ui])
    v.build(vi, v_np[vi])
    I, X = u.eadd(v, api.FLOAT.MINUS).read()
    wi, wx = oracle.v_eadd("FLOAT", u_np[ui], v_np[vi], "MINUS", u_idx=ui, v_idx=vi, u_fill=2.0, v_fill=3.0)
    assert np.array_equal(I, wi) and np.array_equal(X, wx)
    assert u.reduce(api.FLOAT.PLUS, init=1.0) == float(oracle.reduce("FLOAT", u_np[ui], "PLUS", 1.0))
    # dense x dense
    du, dv = api.Vector.from_lists(np.arange(n), u_np, n, api.FLOAT), api.Vector.from_lists(np.arange(n), v_np, n, api.FLOAT)
    du.set_format(1)
    dv.set_format(1)
    r = du.eadd(dv, api.FLOAT.MINUS_POW2)
    assert np.array_equal(r.to_dense(), oracle.v_eadd("FLOAT", u_np, v_np, "MINUS_POW2"))
    assert r.reduce(api.FLOAT.MAX) == float(oracle.reduce("FLOAT", oracle.v_eadd("FLOAT", u_np, v_np, "MINUS_POW2"), "MAX", 0.0))


@pytest.mark.gpu
def test_hip_eadd_reduce_edge_sizes(oracle):
    """sizes around the kernels' internal granules (4-element quads, 256-thread blocks, 1024-element chunks) and empty operands"""
    import ctypes as C
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import binop_id, scalar_bits, type_id
    rng = np.random.default_rng(43)
    t, tid = "INT", type_id("INT")
    for n in (1, 2, 3, 4, 5, 255, 256, 257, 1023, 1024, 1025, 4099):
        x = vals(rng, n, t, 1, 4)
        dx = dev(x)
        for op in ("PLUS", "MAX", "MINUS", "SECOND"):
            out = C.c_uint32()
            _lib.call("splahip_reduce", tid, n, _ptr(dx), scalar_bits(3, tid), C.byref(out), binop_id(op), current_stream_ptr())
            assert np.array([out.value], np.uint32).view(np.int32)[0] == oracle.reduce(t, x, op, 3), (n, op)
        cnt = C.c_uint32()
        _lib.call("splahip_v_count_mf_dense", tid, n, _ptr(dx), scalar_bits(2, tid), C.byref(cnt), current_stream_ptr())
        assert cnt.value == oracle.v_count_mf(t, x, 2)
    # reduce over nothing returns the start value
    out = C.c_uint32()
    _lib.call("splahip_reduce", tid, 0, None, scalar_bits(7, tid), C.byref(out), binop_id("PLUS"), current_stream_ptr())
    assert out.value == 7
    # sparse x sparse with an empty side, disjoint lists, identical lists
    big = 5000
    cases = [(np.array([], np.uint32), np.arange(0, big, 7, dtype=np.uint32)), (np.arange(0, big, 7, dtype=np.uint32), np.array([], np.uint32)),
             (np.arange(0, big, 2, dtype=np.uint32), np.arange(1, big, 2, dtype=np.uint32)), (np.arange(0, big, 3, dtype=np.uint32),) * 2]
    for ui, vi in cases:
        ux, vx = vals(rng, len(ui), t), vals(rng, len(vi), t)
        cap = max(len(ui) + len(vi), 1)
        ri, rx = torch.empty(cap, dtype=torch.int32, device="cuda"), torch.empty(cap, dtype=torch.int32, device="cuda")
        dui, dux, dvi, dvx = (dev(a) if len(a) else None for a in (ui, ux, vi, vx))
        cnt = C.c_uint32()
        _lib.call("splahip_v_eadd_sparse", tid, len(ui), _ptr(dui) if dui is not None else None, _ptr(dux) if dux is not None else None, scalar_bits(2, tid),
                  len(vi), _ptr(dvi) if dvi is not None else None, _ptr(dvx) if dvx is not None else None, scalar_bits(3, tid), _ptr(ri), _ptr(rx), C.byref(cnt),
                  binop_id("MINUS"), current_stream_ptr())
        wi, wx = oracle.v_eadd(t, ux, vx, "MINUS", u_idx=ui, v_idx=vi, u_fill=2, v_fill=3)
        assert cnt.value == len(wi) and np.array_equal(host(ri, np.uint32)[:cnt.value], wi) and np.array_equal(host(rx, np.int32)[:cnt.value], wx)


@pytest.mark.gpu
def test_host_round_trips_of_the_traversal_loops(oracle, monkeypatch):
    """What a traversal level asks the host for, and how: (a) splahip_vec_dense_to_coo_limited returns the count always and the entries only when there are at most
    `limit` (v_count_mf goes through it: reference src/cpu/cpu_v_count_mf.hpp:99-108 counts `!= fill`, the same entries the conversion keeps); (b) uploads of
    <= 16 words travel as kernel arguments and larger ones by copy — same bytes either way; (c) the few words of a host decision arrive the same whether the host
    polls for them or synchronises the stream (SPLAHIP_MAILBOX_POLL_US)."""
    import ctypes as C
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import scalar_bits, type_id
    rng = np.random.default_rng(77)
    for poll in ("100", "0"):
        monkeypatch.setenv("SPLAHIP_MAILBOX_POLL_US", poll)
        for t in ("INT", "FLOAT"):
            tid = type_id(t)
            for n in (1, 5, 1023, 1024, 1025, 70001, 300007):
                x = vals(rng, n, t, 0, 3)
                if n > 100000:                                   # long stretches without a stored entry, and without a gap
                    x[n // 3: n // 2] = 2
                    x[n // 2: n // 2 + 70000] = 1
                keep = np.flatnonzero(x != NP[t](2)).astype(np.uint32)
                dx = dev(x)
                for limit in (0, len(keep) - 1 if len(keep) else 0, len(keep), n):
                    di = torch.full((n,), -1, dtype=torch.int32, device="cuda")
                    dv = torch.full((n,), -1, dtype=torch.int32, device="cuda")
                    cnt = C.c_uint32()
                    _lib.call("splahip_vec_dense_to_coo_limited", tid, n, _ptr(dx), scalar_bits(2, tid), _ptr(di), _ptr(dv), limit, C.byref(cnt), current_stream_ptr())
                    assert cnt.value == len(keep) == oracle.v_count_mf(t, x, 2), (t, n, limit)
                    if len(keep) <= limit:
                        assert np.array_equal(host(di, np.uint32)[:len(keep)], keep) and np.array_equal(host(dv, NP[t])[:len(keep)], x[keep]), (t, n, limit)
                    else:
                        assert bool((di == -1).all()) and bool((dv == -1).all()), (t, n, limit)
    for words in (1, 2, 3, 15, 16, 17, 64):
        src = rng.integers(0, 2**32, size=words, dtype=np.uint64).astype(np.uint32)
        dst = torch.zeros(words + 2, dtype=torch.int32, device="cuda")
        _lib.call("splahip_memcpy_h2d", dst.data_ptr() + 4, src.ctypes.data_as(C.c_void_p), 4 * words, current_stream_ptr())
        src_copy = src.copy()
        src[:] = 0                                      # the caller's memory is its own again on return
        got = host(dst, np.uint32)
        assert got[0] == 0 and got[-1] == 0 and np.array_equal(got[1:-1], src_copy), words
