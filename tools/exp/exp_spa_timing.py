"""Diagnostic: where a workgroup of the blocked numeric form of mxm spends its time (build with SPLA_EXTRA_FLAGS=-DSPLAHIP_SPA_TIMING
SPLA_BUILD_TAG=spatime, run with SPLA_LIB_TAG=spatime).  s_memtime ticks."""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from spla_amd import _lib, graphgen, kernels
torch.cuda.set_device(0); lib = _lib.load(); _lib.call("splahip_init", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
g = torch.Generator(device="cuda"); g.manual_seed(7)
Ax = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float32)
A = (Ap, Aj, Ax)
for _ in range(2):
    R, _info = kernels.mxm("FLOAT", n, n, A, A, "MULT", "PLUS", 0.0, adopt=False)
    del R
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); R, _info = kernels.mxm("FLOAT", n, n, A, A, "MULT", "PLUS", 0.0, adopt=False); b.record(); torch.cuda.synchronize()
dbg = lib.splahip_debug_spa_timing; dbg.argtypes = [C.c_void_p]; dbg.restype = C.c_int
out = np.zeros((4, 1024), np.uint64); dbg(out.ctypes.data)
o = out.astype(np.float64)[:, :256]
tot = o[:3].sum(0)
print(f"mxm call {a.elapsed_time(b):.1f} ms; blocked numeric per workgroup (mean / max ticks): parts+scan {o[0].mean():.0f}/{o[0].max():.0f}  walk {o[1].mean():.0f}/{o[1].max():.0f}  "
      f"emit {o[2].mean():.0f}/{o[2].max():.0f}  steps {o[3].mean():.0f}/{o[3].max():.0f}  total mean {tot.mean():.0f} max {tot.max():.0f} min {tot.min():.0f}; per step {tot.sum() / o[3].sum():.0f} ticks")
