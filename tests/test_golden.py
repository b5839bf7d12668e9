"""Golden vectors captured from the genuine reference CPU backend (tests/golden/make_golden.py):
CPU: the oracle reproduces every case bit for bit; GPU: the HIP path does (fp32 PLUS folds within 1e-5)."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"
NPT = {"INT": np.int32, "UINT": np.uint32, "FLOAT": np.float32}


def _cases():
    manifest = json.loads((GOLD / "hotpath_golden.json").read_text())
    data = np.load(GOLD / "hotpath_golden.npz")
    for c in manifest:
        arrs = {k.split("/", 1)[1]: data[k] for k in data.files if k.startswith(c["name"] + "/")}
        yield c, arrs


def _typed(a, t):
    return a.view(NPT[t])


def test_oracle_reproduces_golden(oracle):
    n = 0
    for c, a in _cases():
        t = c["type"]
        if c["op"] == "mxv":
            r = oracle.mxv_masked(t, c["n_rows"], c["n_cols"], a["Ap"], a["Aj"], _typed(a["Ax"], t), _typed(a["mask"], t), _typed(a["v"], t),
                                  c["mult"], c["add"], c["select"], c["init"], c["early_exit"])
            assert np.array_equal(r.view(np.uint32), a["r"]), c
        elif c["op"] == "vxm":
            ri, rx = oracle.vxm_masked(t, c["n_rows"], c["n_cols"], a["Ap"], a["Aj"], _typed(a["Ax"], t), _typed(a["mask"], t), a["vi"], _typed(a["vx"], t),
                                       c["mult"], c["add"], c["select"])
            assert np.array_equal(ri, a["ri"]) and np.array_equal(rx.view(np.uint32), a["rx"]), c
        elif c["op"] == "mxm":
            R = oracle.mxm(t, c["m"], c["n"], (a["Ap"], a["Aj"], _typed(a["Ax"], t)), (a["Bp"], a["Bj"], _typed(a["Bx"], t)), c["mult"], c["add"], c["init"])
            assert np.array_equal(R[0], a["Rp"]) and np.array_equal(R[1], a["Rj"]) and np.array_equal(R[2].view(np.uint32), a["Rx"]), c
        else:
            R = oracle.mxmT_masked(t, c["m"], (a["Mp"], a["Mj"], _typed(a["Mx"], t)), (a["Ap"], a["Aj"], _typed(a["Ax"], t)),
                                   (a["Bp"], a["Bj"], _typed(a["Bx"], t)), c["mult"], c["add"], c["select"], c["init"])
            assert np.array_equal(R[0], a["Rp"]) and np.array_equal(R[1], a["Rj"]) and np.array_equal(R[2].view(np.uint32), a["Rx"]), c
        n += 1
    assert n == 25


@pytest.mark.gpu
def test_hip_reproduces_golden():
    from gpu_utils import dev, hip_mxv, host
    from spla_amd import kernels

    def close(got, want_bits, t, add):
        want = want_bits.view(NPT[t])
        if t == "FLOAT" and add in ("PLUS", "MULT"):
            np.testing.assert_allclose(got, want, rtol=1e-5)
        else:
            assert np.array_equal(got.view(np.uint32), want_bits)

    def d3(p, j, x, t):
        return dev(p), dev(j), dev(_typed(x, t))

    for c, a in _cases():
        t = c["type"]
        if c["op"] == "mxv":
            r = hip_mxv(t, c["n_rows"], c["n_cols"], a["Ap"], a["Aj"], _typed(a["Ax"], t), _typed(a["mask"], t), _typed(a["v"], t),
                        c["mult"], c["add"], c["select"], c["init"], c["early_exit"])
            close(r, a["r"], t, c["add"] if not c["early_exit"] else "x")
        elif c["op"] == "vxm":
            ri, rx = kernels.vxm_push(t, c["n_rows"], c["n_cols"], *d3(a["Ap"], a["Aj"], a["Ax"], t), dev(_typed(a["mask"], t)), dev(a["vi"]),
                                      dev(_typed(a["vx"], t)), c["mult"], c["add"], c["select"])
            assert np.array_equal(host(ri, np.uint32), a["ri"]) and np.array_equal(host(rx, NPT[t]).view(np.uint32), a["rx"]), c
        elif c["op"] == "mxm":
            (Rp, Rj, Rx), _ = kernels.mxm(t, c["m"], c["n"], d3(a["Ap"], a["Aj"], a["Ax"], t), d3(a["Bp"], a["Bj"], a["Bx"], t), c["mult"], c["add"], c["init"])
            assert np.array_equal(host(Rp, np.uint32), a["Rp"]) and np.array_equal(host(Rj, np.uint32), a["Rj"]), c
            close(host(Rx, NPT[t]), a["Rx"], t, "x")  # runs here are short: folded strictly in order
        else:
            Rp, Rj, Rx = kernels.mxmT_masked(t, c["m"], d3(a["Mp"], a["Mj"], a["Mx"], t), d3(a["Ap"], a["Aj"], a["Ax"], t), d3(a["Bp"], a["Bj"], a["Bx"], t),
                                             c["mult"], c["add"], c["select"], c["init"])
            assert np.array_equal(host(Rp, np.uint32), a["Rp"]) and np.array_equal(host(Rj, np.uint32), a["Rj"]), c
            close(host(Rx, NPT[t]), a["Rx"], t, "x")
