"""Diagnostic: per-phase wave time of spmv_wavetile run items (build with SPLA_EXTRA_FLAGS=-DSPLAHIP_PHASE_TIMING SPLA_BUILD_TAG=phase,
run with SPLA_LIB_TAG=phase).  Clock: s_memtime ticks (100 MHz constant on gfx950 if it matches s_memrealtime; printed both ways)."""
import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from spla_amd import _lib, graphgen, kernels

torch.cuda.set_device(0); lib = _lib.load(); _lib.call("splahip_init", 0)
scale = 22
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
plan = kernels.CsrPlan(n, nnz, Ap)
g = torch.Generator(device="cuda"); g.manual_seed(3)
Ax = torch.rand(nnz, generator=g, device="cuda")
v = torch.rand(n, generator=g, device="cuda")
mask = torch.zeros(n, device="cuda"); r = torch.empty(n, device="cuda")
dbg = lib.splahip_debug_phase; dbg.argtypes = [C.c_void_p, C.c_int]; dbg.restype = C.c_int

def run(name, aj, sel="ALWAYS", steps=10):
    fn = lambda: kernels.mxv_masked(plan, "FLOAT", n, Ap, aj, Ax, mask, v, r, "MULT", "PLUS", sel, 0.0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    out = np.zeros((4, 1 << 18), np.uint32); dbg(out.ctypes.data, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / steps * 1e3
    dbg(out.ctypes.data, 1)
    live = out[3] != 0
    cnt = int(live.sum())
    ph = [float(out[i][live].mean()) for i in range(4)]
    tot = sum(ph)
    print(f"{name:34s} {us:8.1f} us/launch  run items {cnt}  ticks/item: header {ph[0]:8.1f} stream+Ap {ph[1]:8.1f} gather {ph[2]:8.1f} fold {ph[3]:8.1f}  total {tot:8.1f}", flush=True)

k = torch.arange(nnz, device="cuda", dtype=torch.int64)
run("real columns, ALWAYS", Aj)
run("coalesced k % n, ALWAYS", (k % n).to(torch.int32))
run("real columns, EQZERO(all sel)", Aj, "EQZERO")
