#!/usr/bin/env python3
"""Generate tests/golden/hotpath_golden.npz from the GENUINE reference CPU backend (oracle/_ref/libspla_ref.so, built by
oracle/Makefile from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

Each case stores its inputs and the reference's outputs for one of the four hot-path operators over one
(type, op_mult, op_add, op_select, early_exit) cell; the case list covers the reference's own unit-test inputs
(tests/test_mxv.cpp, test_vxm.cpp, test_mxm.cpp, test_mxmT.cpp) and seeded random inputs.  Arrays are stored as uint32
bit patterns so int32 / uint32 / fp32 round-trip exactly.
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from helpers import random_csr, random_mask, random_values  # noqa: E402
from oracle import pyoracle  # noqa: E402


def w(a):
    return np.ascontiguousarray(a).view(np.uint32)


def main():
    ref = pyoracle.Reference()
    rng = np.random.default_rng(2024)
    arrays, manifest = {}, []

    def put(name, **kw):
        for k, v in kw.items():
            arrays[f"{name}/{k}"] = w(np.asarray(v))

    cid = 0
    # ---- mxv
    cells = [("INT", "MULT", "PLUS", "EQZERO", 0, False), ("INT", "BAND", "BOR", "EQZERO", 0, True), ("INT", "PLUS", "MIN", "ALWAYS", 1 << 30, False),
             ("UINT", "LAND", "LOR", "NQZERO", 0, True), ("UINT", "MULT", "BXOR", "GTZERO", 0, False), ("FLOAT", "MULT", "PLUS", "ALWAYS", 0.0, False),
             ("FLOAT", "PLUS", "MIN", "GEZERO", 1e9, False), ("FLOAT", "MULT", "SECOND", "LTZERO", 0.0, False), ("INT", "MULT", "MINUS", "LEZERO", 0, False),
             ("FLOAT", "MULT", "PLUS", "NQZERO", 0.0, True)]
    for (t, mul, add, sel, init, early) in cells:
        n_rows, n_cols = 257, 193
        Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 7, t, skew=True, long_rows=1, long_len=150)
        mask, v = random_mask(rng, n_rows, t), random_values(rng, n_cols, t, zeros=0.25)
        r = ref.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, early)
        name = f"c{cid:03d}"
        cid += 1
        put(name, Ap=Ap, Aj=Aj, Ax=Ax, mask=mask, v=v, r=r)
        manifest.append(dict(name=name, op="mxv", type=t, mult=mul, add=add, select=sel, init=init, early_exit=early, n_rows=n_rows, n_cols=n_cols))
    # the reference's unit case (tests/test_mxv.cpp:33-89)
    Ap, Aj, Ax = np.array([0, 2, 4, 5, 6], np.uint32), np.array([1, 4, 0, 4, 2, 4], np.uint32), np.array([2, -9, 2, -8, 3, -1], np.int32)
    mask, v = np.array([1, 0, 1, 0], np.int32), np.array([3, 0, 3, 0, -1], np.int32)
    r = ref.mxv_masked("INT", 4, 5, Ap, Aj, Ax, mask, v, "MULT", "PLUS", "EQZERO", 0)
    assert r.tolist() == [0, 14, 0, 1]
    name = f"c{cid:03d}"; cid += 1
    put(name, Ap=Ap, Aj=Aj, Ax=Ax, mask=mask, v=v, r=r)
    manifest.append(dict(name=name, op="mxv", type="INT", mult="MULT", add="PLUS", select="EQZERO", init=0, early_exit=False, n_rows=4, n_cols=5, source="tests/test_mxv.cpp:33-89"))
    # ---- vxm
    for (t, mul, add, sel) in [("INT", "MULT", "PLUS", "EQZERO"), ("INT", "BAND", "BOR", "ALWAYS"), ("UINT", "PLUS", "MIN", "NQZERO"),
                               ("FLOAT", "MULT", "PLUS", "ALWAYS"), ("FLOAT", "MULT", "FIRST", "GTZERO"), ("INT", "MULT", "MINUS", "ALWAYS")]:
        n_rows, n_cols = 181, 223
        Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, 6, t, skew=True)
        mask = random_mask(rng, n_cols, t)
        nv = int(rng.integers(5, n_rows))
        vi = rng.permutation(n_rows)[:nv].astype(np.uint32)
        if add not in ("FIRST", "MINUS"):
            vi = np.sort(vi)
        vx = random_values(rng, nv, t, zeros=0.2)
        ri, rx = ref.vxm_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, vi, vx, mul, add, sel)
        name = f"c{cid:03d}"; cid += 1
        put(name, Ap=Ap, Aj=Aj, Ax=Ax, mask=mask, vi=vi, vx=vx, ri=ri, rx=rx)
        manifest.append(dict(name=name, op="vxm", type=t, mult=mul, add=add, select=sel, n_rows=n_rows, n_cols=n_cols))
    # ---- mxm / mxmT
    for (t, mul, add, init, sel) in [("FLOAT", "MULT", "PLUS", 0.0, "GTZERO"), ("INT", "MULT", "PLUS", 0, "ALWAYS"), ("INT", "PLUS", "MIN", 100, "NQZERO"),
                                     ("UINT", "MULT", "SECOND", 0, "EQZERO")]:
        m, k, n = 61, 53, 47
        A, B, Bt = random_csr(rng, m, k, 5, t), random_csr(rng, k, n, 5, t), random_csr(rng, n, k, 5, t)
        Mk = random_csr(rng, m, n, 6, t)
        Mk = (Mk[0], Mk[1], random_mask(rng, len(Mk[1]), t))
        R = ref.mxm(t, m, n, A, B, mul, add, init)
        name = f"c{cid:03d}"; cid += 1
        put(name, Ap=A[0], Aj=A[1], Ax=A[2], Bp=B[0], Bj=B[1], Bx=B[2], Rp=R[0], Rj=R[1], Rx=R[2])
        manifest.append(dict(name=name, op="mxm", type=t, mult=mul, add=add, init=init, m=m, k=k, n=n))
        R = ref.mxmT_masked(t, m, Mk, A, Bt, mul, add, sel, init, n_cols=n, inner=k)
        name = f"c{cid:03d}"; cid += 1
        put(name, Mp=Mk[0], Mj=Mk[1], Mx=Mk[2], Ap=A[0], Aj=A[1], Ax=A[2], Bp=Bt[0], Bj=Bt[1], Bx=Bt[2], Rp=R[0], Rj=R[1], Rx=R[2])
        manifest.append(dict(name=name, op="mxmT", type=t, mult=mul, add=add, select=sel, init=init, m=m, k=k, n=n))

    out = Path(__file__).resolve().parent
    np.savez_compressed(out / "hotpath_golden.npz", **arrays)
    (out / "hotpath_golden.json").write_text(json.dumps(manifest, indent=1))
    print(f"wrote {len(manifest)} cases, {sum(a.nbytes for a in arrays.values()) // 1024} KiB raw")


if __name__ == "__main__":
    main()
