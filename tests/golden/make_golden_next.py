#!/usr/bin/env python3
"""Generate tests/golden/nextrows_golden.{npz,json} from the GENUINE reference CPU backend (oracle/_ref/libspla_ref.so): the
operations next to the hot path — v_assign_masked, v_eadd_fdb, v_count_mf, v_eadd, v_reduce, m_reduce,
m_transpose, v_map — and the four graph algorithms on a small graph.  Run in the build container only:

    python tests/golden/make_golden_next.py

Inputs and the reference's outputs are stored as uint32 bit patterns (int32 / uint32 / fp32 round-trip exactly)."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from helpers import random_csr, random_mask, random_values  # noqa: E402
from oracle import pyoracle  # noqa: E402
from test_algorithms import lower_triangle, pagerank_matrix, sym_graph  # noqa: E402


def w(a):
    return np.ascontiguousarray(a).view(np.uint32)


def main():
    ref = pyoracle.Reference()
    rng = np.random.default_rng(4048)
    arrays, manifest = {}, []
    cid = [0]

    def case(op, arrs, **meta):
        name = f"n{cid[0]:03d}"
        cid[0] += 1
        for k, v in arrs.items():
            arrays[f"{name}/{k}"] = w(np.asarray(v))
        manifest.append(dict(name=name, op=op, **meta))

    n = 311
    for t in ("INT", "UINT", "FLOAT"):
        r, mask, v = random_values(rng, n, t), random_mask(rng, n, t), random_values(rng, n, t, zeros=0.3)
        idx = np.sort(rng.permutation(n)[:97]).astype(np.uint32)
        for op, sel in (("PLUS", "NQZERO"), ("SECOND", "EQZERO"), ("MIN", "GTZERO")):
            case("v_assign_dense", dict(r=r, mask=mask, out=ref.v_assign_masked(t, r, mask, 5, op, sel)), type=t, binop=op, select=sel, value=5)
            case("v_assign_sparse", dict(r=r, idx=idx, mask=mask[idx], out=ref.v_assign_masked(t, r, mask[idx], 5, op, sel, mask_idx=idx)),
                 type=t, binop=op, select=sel, value=5)
        for op in ("PLUS", "MIN", "MINUS"):
            out, fdb = ref.v_eadd_fdb(t, r, v, op, fdb_fill=9)
            case("v_eadd_fdb_dense", dict(r=r, v=v, out=out, fdb=fdb), type=t, binop=op, fdb_fill=9)
            out, (fi, fx) = ref.v_eadd_fdb(t, r, v[idx], op, v_idx=idx)
            case("v_eadd_fdb_sparse", dict(r=r, idx=idx, v=v[idx], out=out, fi=fi, fx=fx), type=t, binop=op)
        case("v_count_mf", dict(v=v), type=t, fill=0, count=int(ref.v_count_mf(t, v, 0)))
        vi = np.sort(rng.permutation(n)[:61]).astype(np.uint32)
        for op in ("PLUS", "MINUS", "MAX"):
            case("v_eadd_dense", dict(u=r, v=v, out=ref.v_eadd(t, r, v, op)), type=t, binop=op)
            ri, rx = ref.v_eadd(t, r[idx], v[vi], op, u_idx=idx, v_idx=vi, u_fill=2, v_fill=3)
            case("v_eadd_sparse", dict(ui=idx, ux=r[idx], vi=vi, vx=v[vi], ri=ri, rx=rx), type=t, binop=op, u_fill=2, v_fill=3, n=n)
        for op, s in (("PLUS", 3), ("MAX", 0), ("MINUS", 100)):
            case("v_reduce", dict(x=v, out=np.array([ref.reduce(t, v, op, s)], pyoracle.TYPE_NP[pyoracle._tid(t)])), type=t, binop=op, s=s)
        Ap, Aj, Ax = random_csr(rng, 83, 127, 5, t, skew=True, long_rows=1, long_len=90)
        for op, s in (("PLUS", 1), ("MIN", 50)):
            case("m_reduce", dict(Ap=Ap, Aj=Aj, Ax=Ax, out=np.array([ref.m_reduce(t, 83, 127, Ap, Aj, Ax, op, s)], pyoracle.TYPE_NP[pyoracle._tid(t)])),
                 type=t, binop=op, s=s, n_rows=83, n_cols=127)
        unops = ["IDENTITY", "AINV", "ABS", "LNOT", "UONE"] + (["SQRT", "FLOOR", "ROUND"] if t == "FLOAT" else ["BNOT"])
        for op in unops:
            x = Ax if op != "SQRT" else np.abs(Ax)
            Rp, Rj, Rx = ref.m_transpose(t, 83, 127, Ap, Aj, x, op)
            case("m_transpose", dict(Ap=Ap, Aj=Aj, Ax=x, Rp=Rp, Rj=Rj, Rx=Rx), type=t, unop=op, n_rows=83, n_cols=127)
            xv = v if op != "SQRT" else np.abs(v)
            case("v_map_dense", dict(x=xv, out=ref.v_map(t, xv, op)), type=t, unop=op)
            mi, mx = ref.v_map(t, xv[idx], op, idx=idx, n=n)
            case("v_map_sparse", dict(idx=idx, x=xv[idx], ri=mi, rx=mx), type=t, unop=op, n=n)

    # ---- the four algorithms on one small symmetric graph (edges both ways, weights for sssp, scaled for pr)
    g = 400
    Ap, Aj, Ax = sym_graph(rng, g, 5, weights=True)
    src = int(np.argmax(np.diff(Ap.astype(np.int64))))
    case("bfs", dict(Ap=Ap, Aj=Aj, out=ref.bfs(g, Ap, Aj, src)), n=g, source=src)
    case("sssp", dict(Ap=Ap, Aj=Aj, Ax=Ax, out=ref.sssp(g, Ap, Aj, Ax, src)), n=g, source=src)
    Px = pagerank_matrix(Ap, Aj)
    case("pr", dict(Ap=Ap, Aj=Aj, Ax=Px, out=ref.pr(g, Ap, Aj, Px, 0.85, 1e-6)), n=g, alpha=0.85, eps=1e-6)
    Lp, Lj = lower_triangle(Ap, Aj)
    count, _ = ref.tc(g, Lp, Lj)
    case("tc", dict(Lp=Lp, Lj=Lj), n=g, triangles=int(count))

    out = Path(__file__).resolve().parent
    np.savez_compressed(out / "nextrows_golden.npz", **arrays)
    (out / "nextrows_golden.json").write_text(json.dumps(manifest, indent=1))
    print(f"wrote {len(manifest)} cases, {sum(a.nbytes for a in arrays.values()) // 1024} KiB raw")


if __name__ == "__main__":
    main()
