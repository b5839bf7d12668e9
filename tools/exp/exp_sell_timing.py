"""Diagnostic: where a wave of the sliced-ELL bit-form kernel spends its time on the BFS step (build with
SPLA_EXTRA_FLAGS=-DSPLAHIP_SELL_TIMING SPLA_BUILD_TAG=selltime, run with SPLA_LIB_TAG=selltime).  s_memtime ticks (100 MHz)."""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from spla_amd import _lib, graphgen, kernels
torch.cuda.set_device(0); lib = _lib.load(); _lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(22, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
Ax = torch.ones(nnz, dtype=torch.int32, device="cuda")
front = Aj[int(Ap[0]):int(Ap[1])].to(torch.int64)
v = torch.zeros(n, dtype=torch.int32, device="cuda"); v[front] = 1
depth = torch.zeros(n, dtype=torch.int32, device="cuda"); depth[front] = 2; depth[0] = 1
sell = kernels.CsrPlan(n, nnz, Ap, n, Aj, Ax)
r = torch.empty(n, dtype=torch.int32, device="cuda")
dbg = lib.splahip_debug_sell_timing; dbg.argtypes = [C.c_void_p]; dbg.restype = C.c_int
for name, sel in (("A bfs step", "EQZERO"), ("B all rows", "ALWAYS")):
    fn = lambda: kernels.mxv_masked(sell, "INT", n, Ap, Aj, Ax, depth, v, r, "BAND", "BOR", sel, 0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    out = np.zeros((12, 4096), np.uint64); dbg(out.ctypes.data)
    o = out.astype(np.float64)
    print(f"{name}: call {a.elapsed_time(b) * 1e3:.1f} us (with stamps); per wave (mean / max), ticks: select+request {o[0].mean():.0f}/{o[0].max():.0f}  "
          f"loads+walk {o[1].mean():.0f}/{o[1].max():.0f}  result {o[2].mean():.0f}/{o[2].max():.0f}  slices {o[3].mean():.1f}/{o[3].max():.0f}  "
          f"wave total {o[4].mean():.0f}/{o[4].max():.0f} min {o[4].min():.0f}", flush=True)
    print(f"    per wave: live slices {o[6].mean():.1f}/{o[6].max():.0f}  group rounds {o[7].mean():.1f}/{o[7].max():.0f}  sum of widths {o[8].mean():.0f}/{o[8].max():.0f}  "
          f"selected lanes per live slice {o[9].sum() / max(o[6].sum(), 1):.1f}  entry..first slice {o[10].mean():.0f}/{o[10].max():.0f} ticks; "
          f"chip: live slices {o[6].sum():.0f} rounds {o[7].sum():.0f}")
    per = o[:3].sum(0) / np.maximum(o[3], 1)
    print(f"    per slice: {per.mean():.0f} ticks; waves by total: p10 {np.percentile(o[4], 10):.0f} p50 {np.percentile(o[4], 50):.0f} p90 {np.percentile(o[4], 90):.0f}")
