"""Randomised soak of the masked SpMV forms (CSR kernels, sliced-ELL auto, sliced-ELL bit form forced, column-ordered blocks in two geometries; one to three
calls per plan, so that the selected-rows kernels and the first-hit kernel — which follow the hint of an earlier call — are reached) against the oracle:
random shapes, degree skews, long rows, semirings, select ops, init values and input-vector densities for a given time budget.
    python tools/exp/fuzz_mxv.py [seconds] [seed]"""
import os, sys, time
os.environ["SPLA_ENV_AUTOSYNC"] = "1"  # knob changes made by the test helpers mid-process take effect (spla_amd/_lib.py: _sync_env)
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from helpers import NP, SELECTS, SEMIRINGS_COMMON, SEMIRINGS_INT, assert_same, random_csr, random_mask, random_values
from gpu_utils import hip_mxv
from oracle import pyoracle
from spla_amd import _lib
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
orc = pyoracle.Oracle()
t0, n = time.time(), 0
while time.time() - t0 < budget:
    t = ["INT", "UINT", "FLOAT"][int(rng.integers(0, 3))]
    n_rows, n_cols = int(rng.integers(1, 6000)), int(rng.integers(1, 60000))
    long_rows = int(rng.integers(0, 4))
    long_len = int(min(n_cols, rng.choice([257, 300, 700, 5000, 16384, 16500, 33000])))
    Ap, Aj, Ax = random_csr(rng, n_rows, n_cols, float(rng.choice([0.5, 3, 12, 40])), t, skew=bool(rng.integers(0, 2)), empty_frac=float(rng.choice([0.0, 0.3, 0.9])),
                            long_rows=long_rows, long_len=long_len)
    if rng.random() < 0.2 and len(Ax):  # a matrix of equal values: the structure-only pass of the sliced-ELL bit form
        Ax = np.full_like(Ax, Ax[0] if rng.integers(0, 2) else 1)
    sems = SEMIRINGS_COMMON + (SEMIRINGS_INT if t != "FLOAT" else []) + ([("MULT", "BXOR"), ("LAND", "PLUS"), ("BAND", "BOR")] if t != "FLOAT" else [])
    mul, add = sems[int(rng.integers(0, len(sems)))]
    sel = SELECTS[int(rng.integers(0, len(SELECTS)))]
    init = int(rng.choice([0, 1, 5, 100]))
    dens = float(rng.choice([0.0, 0.01, 0.2, 1.0]))
    v = random_values(rng, n_cols, t)
    if dens < 1.0:
        v = np.where(rng.random(n_cols) < dens, v if rng.integers(0, 2) else np.ones_like(v), 0).astype(NP[t])
    mask = random_mask(rng, n_rows, t)
    want = orc.mxv_masked(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init)
    # every third case: the selected-rows kernels forced on whatever the mask (round 6: short rows in batches, medium rows and long rows' segments a wave each)
    if rng.random() < 0.34:
        os.environ["SPLAHIP_MASK_ROWS_X10"], os.environ["SPLAHIP_MASK_LONG_X10"] = "1", "1"
    else:
        os.environ.pop("SPLAHIP_MASK_ROWS_X10", None), os.environ.pop("SPLAHIP_MASK_LONG_X10", None)
    try:
        got = hip_mxv(t, n_rows, n_cols, Ap, Aj, Ax, mask, v, mul, add, sel, init, junk=-3, calls=int(rng.integers(1, 4)))
    except AssertionError as e:
        print(f"FAILED case {n}: {t} {n_rows}x{n_cols} nnz={len(Aj)} {mul}/{add}/{sel} init={init} dens={dens} long={long_rows}x{long_len}: {e}")
        np.savez("/tmp/fuzz_fail.npz", Ap=Ap, Aj=Aj, Ax=Ax, mask=mask, v=v)
        import os
        from gpu_utils import dev, host
        from spla_amd import kernels
        for form in ("csr", "sell"):
            os.environ["SPLAHIP_SELL_BITS"] = "1"
            dAp, dAj, dAx, dm, dv = dev(Ap), dev(Aj), dev(np.asarray(Ax, NP[t])), dev(np.asarray(mask, NP[t])), dev(np.asarray(v, NP[t]))
            plan = kernels.CsrPlan(n_rows, len(Aj), dAp, n_cols, dAj, dAx) if form == "sell" else kernels.CsrPlan(n_rows, len(Aj), dAp)
            r = torch.empty(n_rows, dtype=kernels.torch_dtype(t), device="cuda")
            kernels.mxv_masked(plan, t, n_cols, dAp, dAj, dAx, dm, dv, r, mul, add, sel, init, False)
            torch.cuda.synchronize()
            g = host(r, NP[t])
            bad = np.nonzero(~((g == want) | (np.isnan(g) & np.isnan(want))))[0]
            print(form, "rows differing from the oracle:", len(bad), bad[:8], g[bad[:4]], want[bad[:4]], "row lens", np.diff(Ap.astype(np.int64))[bad[:8]])
        raise
    assert_same(got, want, t, what=f"case {n}: {t} {n_rows}x{n_cols} nnz={len(Aj)} {mul}/{add}/{sel} init={init} dens={dens} long={long_rows}x{long_len}")
    n += 1
print(f"fuzz_mxv: {n} random cases passed in {time.time() - t0:.0f} s (seed {seed})")
