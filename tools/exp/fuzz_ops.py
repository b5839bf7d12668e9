"""Randomised soak of vxm (push, small-frontier push, pull, pull over the sliced-ELL copy), mxm and mxmT against the oracle.
    python tools/exp/fuzz_ops.py [seconds] [seed]"""
import os, sys, time
os.environ["SPLA_ENV_AUTOSYNC"] = "1"  # knob changes made by the test helpers mid-process take effect (spla_amd/_lib.py: _sync_env)
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from helpers import NP, SELECTS, SEMIRINGS_COMMON, SEMIRINGS_INT, SEMIRINGS_ORDERED, random_csr, random_mask, random_values
from gpu_utils import dev, host
from oracle import pyoracle
from spla_amd import _lib, kernels
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
rng = np.random.default_rng(seed)
orc = pyoracle.Oracle()
ORDER_FREE = {"MIN", "MAX", "LOR", "LAND", "BOR", "BAND", "BXOR"}


def close(t, add, got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if t == "FLOAT" and add in ("PLUS",):
        assert np.allclose(got, want, rtol=1e-5, atol=0, equal_nan=True), what
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what


def d3(M, t):
    return dev(M[0]), dev(M[1]), dev(np.asarray(M[2], NP[t]))


t0, n, counts = time.time(), 0, {}
while time.time() - t0 < budget:
    t = ["INT", "UINT", "FLOAT"][int(rng.integers(0, 3))]
    sems = SEMIRINGS_COMMON + SEMIRINGS_ORDERED + (SEMIRINGS_INT if t != "FLOAT" else [])
    mul, add = sems[int(rng.integers(0, len(sems)))]
    sel = SELECTS[int(rng.integers(0, len(SELECTS)))]
    kind = ["vxm", "vxm", "mxm", "mxmT"][int(rng.integers(0, 4))]
    what = f"case {n}: {kind} {t} {mul}/{add}/{sel}"
    if kind == "vxm":
        n_rows, n_cols = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
        A = random_csr(rng, n_rows, n_cols, float(rng.choice([1, 6, 25])), t, skew=bool(rng.integers(0, 2)), long_rows=int(rng.integers(0, 3)),
                       long_len=int(min(n_cols, rng.choice([100, 700, 2500]))))
        uniform = rng.random() < 0.3 and len(A[2]) > 0  # a structure-only matrix: the pull form may settle an output entry at its first stored in-neighbour
        if uniform:
            A = (A[0], A[1], np.full_like(A[2], A[2][0]))
        mask = random_mask(rng, n_cols, t)
        nv = int(rng.choice([1, max(1, n_rows // 50), max(1, n_rows // 3), n_rows]))
        vi = np.sort(rng.permutation(n_rows)[:nv]).astype(np.uint32)
        vx = random_values(rng, nv, t, zeros=0.2)
        if uniform and rng.random() < 0.7:
            vx = np.full_like(vx, vx[0])  # a frontier: one stored word (possibly 0)
        want = orc.vxm_masked(t, n_rows, n_cols, *A, mask, vi, vx, mul, add, sel)
        dA, dm, dvi, dvx = d3(A, t), dev(np.asarray(mask, NP[t])), dev(vi), dev(np.asarray(vx, NP[t]))
        ri, rx = kernels.vxm_push(t, n_rows, n_cols, *dA, dm, dvi, dvx, mul, add, sel)
        assert np.array_equal(host(ri, np.uint32), want[0]), what + " push idx"
        assert np.array_equal(host(rx, NP[t]).view(np.uint32), want[1].view(np.uint32)), what + " push val"   # the push form keeps the reference's order: exact
        if add in ORDER_FREE:
            done, _, _, ri, rx = kernels.vxm_push_small(t, n_rows, n_cols, *dA, dm, dvi, dvx, mul, add, sel, 1 << 22)
            assert done and np.array_equal(host(ri, np.uint32), want[0]) and np.array_equal(host(rx, NP[t]).view(np.uint32), want[1].view(np.uint32)), what + " small"
        if add in ORDER_FREE or (add == "PLUS") or (add == "MULT" and t != "FLOAT"):
            Tp, Tj, Tx = kernels.csr_transpose(n_rows, n_cols, *dA)
            for sell in (False, True):
                plan = kernels.CsrPlan(n_cols, len(A[1]), Tp, n_rows, Tj, Tx) if sell else kernels.CsrPlan(n_cols, len(A[1]), Tp)
                if uniform:  # what the owner of the matrix and the caller's probe would say (spla_hip.h: set_uniform_values, hint_products)
                    plan.set_uniform_values(True, int(np.asarray(A[2][:1], NP[t]).view(np.uint32)[0]))
                    plan.hint_products(int(np.diff(A[0].astype(np.int64))[vi].sum()) * int(rng.choice([1, 1000])))
                ri, rx = kernels.vxm_pull(plan, t, n_rows, n_cols, Tp, Tj, Tx, dm, dvi, dvx, mul, add, sel, identity=(nv == n_rows))
                assert np.array_equal(host(ri, np.uint32), want[0]), what + f" pull idx sell={sell}"
                close(t, add, host(rx, NP[t]), want[1], what + f" pull val sell={sell}")
                plan.close()
    else:
        m, k, nn = int(rng.integers(1, 400)), int(rng.integers(1, 400)), int(rng.integers(1, 400))
        deg = float(rng.choice([1, 5, 30]))
        init = int(rng.choice([0, 1, 7]))
        A = random_csr(rng, m, k, deg, t, skew=bool(rng.integers(0, 2)), long_rows=1, long_len=min(k, 300))
        if kind == "mxm":
            B = random_csr(rng, k, nn, deg, t, skew=bool(rng.integers(0, 2)))
            want = orc.mxm(t, m, nn, A, B, mul, add, init)
            (Rp, Rj, Rx), _ = kernels.mxm(t, m, nn, d3(A, t), d3(B, t), mul, add, init)
        else:
            B = random_csr(rng, nn, k, deg, t, skew=bool(rng.integers(0, 2)), long_rows=1, long_len=min(k, 300))
            Mk = random_csr(rng, m, nn, 8, t)
            Mk = (Mk[0], Mk[1], random_mask(rng, len(Mk[1]), t))
            want = orc.mxmT_masked(t, m, Mk, A, B, mul, add, sel, init)
            Rp, Rj, Rx = kernels.mxmT_masked(t, m, d3(Mk, t), d3(A, t), d3(B, t), mul, add, sel, init)
        assert np.array_equal(host(Rp, np.uint32), want[0]) and np.array_equal(host(Rj, np.uint32), want[1]), what + " structure"
        close(t, add, host(Rx, NP[t]), want[2], what + " values")
    counts[kind] = counts.get(kind, 0) + 1
    n += 1
print(f"fuzz_ops: {n} random cases passed in {time.time() - t0:.0f} s (seed {seed}): {counts}")
