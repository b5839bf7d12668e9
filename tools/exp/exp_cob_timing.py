"""Diagnostic: per-workgroup timeline of the column-ordered-blocks SpMV (fp32 plus-times, all rows) — who starts when, on which XCD, how long a
piece runs per chunk, where the tail is.  Build with SPLA_EXTRA_FLAGS=-DSPLAHIP_COB_TIMING SPLA_BUILD_TAG=cobtime, run with SPLA_LIB_TAG=cobtime.
s_memrealtime ticks (100 MHz)."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch

from spla_amd import _lib, graphgen, kernels

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
torch.cuda.set_device(0)
lib = _lib.load()
_lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
gen = torch.Generator(device="cuda")
gen.manual_seed(101)
Ax = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float32)
v = torch.rand(n, generator=gen, device="cuda", dtype=torch.float32)
mask = torch.zeros(n, dtype=torch.float32, device="cuda")
r = torch.empty(n, dtype=torch.float32, device="cuda")
p = kernels.CsrPlan(n, nnz, Ap)
info = p.attach_cob(n, Ap, Aj, Ax, 0, 0)
print(f"R-MAT {scale}: n={n} nnz={nnz} cob {info}")
fn = lambda: kernels.mxv_masked(p, "FLOAT", n, Ap, Aj, Ax, mask, v, r, "MULT", "PLUS", "ALWAYS", 0.0)
for _ in range(5):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); fn(); b.record(); torch.cuda.synchronize()
dbg = lib.splahip_debug_cob_timing
dbg.argtypes = [C.c_void_p]
dbg.restype = C.c_int
out = np.zeros((8, 8192), np.uint64)
dbg(out.ctypes.data)
nt = int((out[0] != 0).sum())
o = out[:, :nt].astype(np.int64)
t0 = o[0].min()
start, zeroed, looped, end = (o[k] - t0 for k in range(4))
xcc, chunks, slab, rng_id = o[4], o[6], o[7] & 0xFFFFFFFF, o[7] >> 32
tick = 0.01  # us
print(f"call {a.elapsed_time(b) * 1e3:.1f} us (with stamps); workgroups {nt}; kernel span {end.max() * tick:.1f} us")
print(f"  starts: first 256 within {np.sort(start)[min(255, nt - 1)] * tick:.2f} us; last start {start.max() * tick:.1f} us")
dur = (end - start) * tick
loop = (looped - zeroed) * tick
print(f"  per piece: zero {np.mean((zeroed - start) * tick):.2f} us, loop {loop.mean():.1f} (max {loop.max():.1f}), flush {np.mean((end - looped) * tick):.2f} (max {np.max((end - looped) * tick):.2f}) us")
big = chunks > np.percentile(chunks, 50)
print(f"  loop ns per chunk: big pieces {1e3 * loop[big].sum() / chunks[big].sum():.1f}, small {1e3 * loop[~big].sum() / max(chunks[~big].sum(), 1):.1f}")
print(f"  ends: p10 {np.percentile(end, 10) * tick:.1f} p50 {np.percentile(end, 50) * tick:.1f} p90 {np.percentile(end, 90) * tick:.1f} max {end.max() * tick:.1f} us")
# busy CUs over time
T = int(end.max()) + 1
ev = np.zeros(T + 1, np.int64)
np.add.at(ev, start, 1)
np.add.at(ev, end, -1)
act = np.cumsum(ev)[:T]
for q in range(10):
    seg = act[T * q // 10: T * (q + 1) // 10]
    print(f"    t {T * q // 10 * tick:6.1f} .. {T * (q + 1) // 10 * tick:6.1f} us: active workgroups mean {seg.mean():6.1f}")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCD {x}: pieces {m.sum():4d} chunks {chunks[m].sum():7d} last end {end[m].max() * tick:6.1f} us  ns/chunk {1e3 * loop[m].sum() / chunks[m].sum():.1f}  blockIdx%8 {np.bincount(np.nonzero(m)[0] % 8, minlength=8).tolist()}")
# chunk-rate of a piece vs its start time: does a piece that runs beside fewer others go faster?
order = np.argsort(start)
for lo in range(0, nt, max(1, nt // 8)):
    sl = order[lo: lo + max(1, nt // 8)]
    print(f"    pieces {lo:4d}..: start {start[sl].mean() * tick:6.1f} us  chunks {chunks[sl].mean():7.1f}  ns/chunk {1e3 * loop[sl].sum() / chunks[sl].sum():6.1f}  slabbed {np.mean(slab[sl] != 0xFFFFFFFF):.2f}")

print("  slowest pieces (end time): ")
for t in np.argsort(-end)[:12]:
    print(f"    piece {t:4d} range {rng_id[t]:4d}{' (multi)' if slab[t] != 0xFFFFFFFF else '        '} chunks {chunks[t]:6d}  start {start[t] * tick:6.1f}  zero {(zeroed[t] - start[t]) * tick:5.2f}  loop {loop[t]:6.1f} "
          f"({1e3 * loop[t] / max(chunks[t], 1):6.1f} ns/chunk)  flush {(end[t] - looped[t]) * tick:6.2f}  end {end[t] * tick:6.1f} us")
fl = (end - looped) * tick
print(f"  flush: p50 {np.percentile(fl, 50):.2f} p90 {np.percentile(fl, 90):.2f} max {fl.max():.2f} us; sum over pieces of (zero + flush) {((zeroed - start) * tick).sum() + fl.sum():.0f} us-CU of {dur.sum():.0f}")
