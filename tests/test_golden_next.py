"""Golden vectors of the operations next to the hot path, captured from the genuine reference CPU backend
by tests/golden/make_golden_next.py.  CPU: the oracle reproduces every case bit for bit.  GPU: the HIP kernels (kernel-level C ABI)
and the device-resident graph algorithms (operator API) do — fp32 PLUS folds and PageRank within the tolerance stated at the check."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"
NPT = {"INT": np.int32, "UINT": np.uint32, "FLOAT": np.float32}


def _cases():
    manifest = json.loads((GOLD / "nextrows_golden.json").read_text())
    data = np.load(GOLD / "nextrows_golden.npz")
    for c in manifest:
        yield c, {k.split("/", 1)[1]: data[k] for k in data.files if k.startswith(c["name"] + "/")}


def _t(a, t):
    return a.view(NPT[t])


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_fixture_covers_every_next_row():
    ops = {c["op"] for c, _ in _cases()}
    assert ops == {"v_assign_dense", "v_assign_sparse", "v_eadd_fdb_dense", "v_eadd_fdb_sparse", "v_count_mf", "v_eadd_dense", "v_eadd_sparse",
                   "v_reduce", "m_reduce", "m_transpose", "v_map_dense", "v_map_sparse", "bfs", "sssp", "pr", "tc"}


def test_oracle_reproduces_golden(oracle):
    n = 0
    for c, a in _cases():
        op, t = c["op"], c.get("type")
        if op == "v_assign_dense":
            assert np.array_equal(_bits(oracle.v_assign_masked(t, _t(a["r"], t), _t(a["mask"], t), c["value"], c["binop"], c["select"])), a["out"]), c
        elif op == "v_assign_sparse":
            got = oracle.v_assign_masked(t, _t(a["r"], t), _t(a["mask"], t), c["value"], c["binop"], c["select"], mask_idx=a["idx"])
            assert np.array_equal(_bits(got), a["out"]), c
        elif op == "v_eadd_fdb_dense":
            out, fdb = oracle.v_eadd_fdb(t, _t(a["r"], t), _t(a["v"], t), c["binop"], fdb_fill=c["fdb_fill"])
            assert np.array_equal(_bits(out), a["out"]) and np.array_equal(_bits(fdb), a["fdb"]), c
        elif op == "v_eadd_fdb_sparse":
            out, (fi, fx) = oracle.v_eadd_fdb(t, _t(a["r"], t), _t(a["v"], t), c["binop"], v_idx=a["idx"])
            assert np.array_equal(_bits(out), a["out"]) and np.array_equal(fi, a["fi"]) and np.array_equal(_bits(fx), a["fx"]), c
        elif op == "v_count_mf":
            assert oracle.v_count_mf(t, _t(a["v"], t), c["fill"]) == c["count"], c
        elif op == "v_eadd_dense":
            assert np.array_equal(_bits(oracle.v_eadd(t, _t(a["u"], t), _t(a["v"], t), c["binop"])), a["out"]), c
        elif op == "v_eadd_sparse":
            ri, rx = oracle.v_eadd(t, _t(a["ux"], t), _t(a["vx"], t), c["binop"], u_idx=a["ui"], v_idx=a["vi"], u_fill=c["u_fill"], v_fill=c["v_fill"])
            assert np.array_equal(ri, a["ri"]) and np.array_equal(_bits(rx), a["rx"]), c
        elif op == "v_reduce":
            assert _bits(np.array([oracle.reduce(t, _t(a["x"], t), c["binop"], c["s"])]))[0] == a["out"][0], c
        elif op == "m_reduce":
            assert _bits(np.array([oracle.reduce(t, _t(a["Ax"], t), c["binop"], c["s"])]))[0] == a["out"][0], c
        elif op == "m_transpose":
            Rp, Rj, Rx = oracle.m_transpose(t, c["n_rows"], c["n_cols"], a["Ap"], a["Aj"], _t(a["Ax"], t), c["unop"])
            assert np.array_equal(Rp, a["Rp"]) and np.array_equal(Rj, a["Rj"]) and np.array_equal(_bits(Rx), a["Rx"]), c
        elif op == "v_map_dense":
            assert np.array_equal(_bits(oracle.v_map(t, _t(a["x"], t), c["unop"])), a["out"]), c
        elif op == "v_map_sparse":
            assert np.array_equal(_bits(oracle.v_map(t, _t(a["x"], t), c["unop"])), a["rx"]) and np.array_equal(a["idx"], a["ri"]), c
        elif op == "bfs":
            assert np.array_equal(oracle.bfs(c["n"], a["Ap"], a["Aj"], c["source"]), a["out"].view(np.int32)), c
        elif op == "sssp":
            assert np.array_equal(_bits(oracle.sssp(c["n"], a["Ap"], a["Aj"], _t(a["Ax"], "FLOAT"), c["source"])), a["out"]), c
        elif op == "pr":
            got = oracle.pr(c["n"], a["Ap"], a["Aj"], _t(a["Ax"], "FLOAT"), c["alpha"], c["eps"])
            assert np.array_equal(_bits(got), a["out"]), c
        elif op == "tc":
            assert oracle.tc(c["n"], a["Lp"], a["Lj"])[0] == c["triangles"], c
        n += 1
    assert n == 136


@pytest.mark.gpu
def test_hip_reproduces_golden():
    import torch
    from gpu_utils import dev, host
    from spla_amd import _lib, api
    from spla_amd.kernels import _ptr, current_stream_ptr
    from spla_amd.ops import binop_id, scalar_bits, selop_id, type_id, unop_id
    S = current_stream_ptr

    def eq(t_dev, want_bits, n=None):
        got = host(t_dev, np.uint32) if t_dev.dtype == torch.int32 else host(t_dev, np.float32).view(np.uint32)
        return np.array_equal(got[:len(want_bits) if n is None else n], want_bits)

    for c, a in _cases():
        op, t = c["op"], c.get("type")
        tid = type_id(t) if t else None
        if op == "v_assign_dense":
            dr, dm = dev(_t(a["r"], t).copy()), dev(_t(a["mask"], t))
            _lib.call("splahip_v_assign_masked_dense", tid, len(a["r"]), _ptr(dr), _ptr(dm), scalar_bits(c["value"], tid), binop_id(c["binop"]), selop_id(c["select"]), S())
            assert eq(dr, a["out"]), c
        elif op == "v_assign_sparse":
            dr, di, dm = dev(_t(a["r"], t).copy()), dev(a["idx"]), dev(_t(a["mask"], t))
            _lib.call("splahip_v_assign_masked_sparse", tid, len(a["idx"]), _ptr(dr), _ptr(di), _ptr(dm), scalar_bits(c["value"], tid), binop_id(c["binop"]),
                      selop_id(c["select"]), S())
            assert eq(dr, a["out"]), c
        elif op == "v_eadd_fdb_dense":
            dr, dv = dev(_t(a["r"], t).copy()), dev(_t(a["v"], t))
            df = torch.empty_like(dr)
            _lib.call("splahip_v_eadd_fdb_dense", tid, len(a["r"]), _ptr(dr), _ptr(dv), _ptr(df), scalar_bits(c["fdb_fill"], tid), binop_id(c["binop"]), S())
            assert eq(dr, a["out"]) and eq(df, a["fdb"]), c
        elif op == "v_eadd_fdb_sparse":
            dr, di, dv = dev(_t(a["r"], t).copy()), dev(a["idx"]), dev(_t(a["v"], t))
            fi, fx = torch.empty(len(a["idx"]), dtype=torch.int32, device="cuda"), torch.empty(len(a["idx"]), dtype=dr.dtype, device="cuda")
            cnt = C.c_uint32()
            _lib.call("splahip_v_eadd_fdb_sparse", tid, len(a["idx"]), _ptr(dr), _ptr(di), _ptr(dv), _ptr(fi), _ptr(fx), C.byref(cnt), binop_id(c["binop"]), S())
            assert eq(dr, a["out"]) and cnt.value == len(a["fi"]) and eq(fi, a["fi"]) and eq(fx, a["fx"]), c
        elif op == "v_count_mf":
            dv, cnt = dev(_t(a["v"], t)), C.c_uint32()
            _lib.call("splahip_v_count_mf_dense", tid, len(a["v"]), _ptr(dv), scalar_bits(c["fill"], tid), C.byref(cnt), S())
            assert cnt.value == c["count"], c
        elif op == "v_eadd_dense":
            du, dv = dev(_t(a["u"], t)), dev(_t(a["v"], t))
            dr = torch.empty_like(du)
            _lib.call("splahip_v_eadd_dense", tid, len(a["u"]), _ptr(dr), _ptr(du), _ptr(dv), binop_id(c["binop"]), S())
            assert eq(dr, a["out"]), c
        elif op == "v_eadd_sparse":
            dui, dux, dvi, dvx = dev(a["ui"]), dev(_t(a["ux"], t)), dev(a["vi"]), dev(_t(a["vx"], t))
            cap = len(a["ui"]) + len(a["vi"])
            ri, rx = torch.empty(cap, dtype=torch.int32, device="cuda"), torch.empty(cap, dtype=dux.dtype, device="cuda")
            cnt = C.c_uint32()
            _lib.call("splahip_v_eadd_sparse", tid, len(a["ui"]), _ptr(dui), _ptr(dux), scalar_bits(c["u_fill"], tid), len(a["vi"]), _ptr(dvi), _ptr(dvx),
                      scalar_bits(c["v_fill"], tid), _ptr(ri), _ptr(rx), C.byref(cnt), binop_id(c["binop"]), S())
            assert cnt.value == len(a["ri"]) and eq(ri, a["ri"]) and eq(rx, a["rx"]), c
        elif op in ("v_reduce", "m_reduce"):
            x = a["x"] if op == "v_reduce" else a["Ax"]
            dx, out = dev(_t(x, t)), C.c_uint32()
            _lib.call("splahip_reduce", tid, len(x), _ptr(dx), scalar_bits(c["s"], tid), C.byref(out), binop_id(c["binop"]), S())
            if t == "FLOAT" and c["binop"] == "PLUS":  # reduction tree vs the reference's sequential fold: rounding only
                np.testing.assert_allclose(np.array([out.value], np.uint32).view(np.float32), a["out"].view(np.float32), rtol=1e-5)
            else:
                assert out.value == a["out"][0], c
        elif op == "m_transpose":
            nnz = len(a["Aj"])
            dAp, dAj, dAx = dev(a["Ap"]), dev(a["Aj"]), dev(_t(a["Ax"], t))
            Rp = torch.empty(c["n_cols"] + 1, dtype=torch.int32, device="cuda")
            Rj, Rx = torch.empty(nnz, dtype=torch.int32, device="cuda"), torch.empty(nnz, dtype=torch.int32, device="cuda")
            _lib.call("splahip_m_transpose", tid, c["n_rows"], c["n_cols"], nnz, _ptr(dAp), _ptr(dAj), _ptr(dAx), _ptr(Rp), _ptr(Rj), _ptr(Rx), unop_id(c["unop"]), S())
            torch.cuda.synchronize()
            assert eq(Rp, a["Rp"]) and eq(Rj, a["Rj"]) and eq(Rx, a["Rx"]), c
        elif op in ("v_map_dense", "v_map_sparse"):
            dx = dev(_t(a["x"], t))
            out = torch.empty_like(dx)
            _lib.call("splahip_map", tid, len(a["x"]), _ptr(dx), _ptr(out), unop_id(c["unop"]), S())
            assert eq(out, a["out"] if op == "v_map_dense" else a["rx"]), c
        elif op in ("bfs", "sssp", "pr", "tc"):
            n = c["n"]
            if op == "tc":
                LI = np.repeat(np.arange(n, dtype=np.uint32), np.diff(a["Lp"].astype(np.int64)))
                count, _ = api.tc(api.Matrix.from_lists(LI, a["Lj"], np.ones(len(a["Lj"]), np.int32), (n, n), api.INT))
                assert count == c["triangles"], c
                continue
            I = np.repeat(np.arange(n, dtype=np.uint32), np.diff(a["Ap"].astype(np.int64)))
            if op == "bfs":
                got = api.bfs(api.Matrix.from_lists(I, a["Aj"], np.ones(len(a["Aj"]), np.int32), (n, n), api.INT), c["source"]).to_dense()
                assert np.array_equal(got, a["out"].view(np.int32)), c
            elif op == "sssp":
                got = api.sssp(api.Matrix.from_lists(I, a["Aj"], _t(a["Ax"], "FLOAT"), (n, n), api.FLOAT), c["source"]).to_dense()
                assert np.array_equal(_bits(got), a["out"]), c  # min-plus over fp32 sums of two terms: exact
            else:
                got = api.pr(api.Matrix.from_lists(I, a["Aj"], _t(a["Ax"], "FLOAT"), (n, n), api.FLOAT), c["alpha"], c["eps"]).to_dense()
                np.testing.assert_allclose(got, a["out"].view(np.float32), rtol=1e-4, atol=1e-7)  # both within eps of the fixed point
