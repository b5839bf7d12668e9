"""Diagnostic: per-visit timeline of the run-time schedule of the blocks kernel (mxv_cob_dyn.hpp): one record per (workgroup, range) visit.
Build with SPLA_EXTRA_FLAGS=-DSPLAHIP_COB_TIMING SPLA_BUILD_TAG=cobtime, run with SPLA_LIB_TAG=cobtime.  s_memrealtime ticks (100 MHz)."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch

from spla_amd import _lib, graphgen, kernels

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
for kv in sys.argv[2:]:
    k, v_ = kv.split("=", 1)
    os.environ[k] = v_
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
lib.splahip_debug_cob_timing_reset.restype = C.c_int
lib.splahip_debug_cob_timing_reset()
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
xcc, wg, stolen, joined, chunks, rng_id = o[4], o[5] & 0xFFFFFFFF, (o[5] >> 32) & 1, (o[5] >> 33) & 1, o[6], o[7] >> 32
tick = 0.01
print(f"call {a.elapsed_time(b) * 1e3:.1f} us (with stamps); visits {nt} ({int(joined.sum())} joined, {int(stolen.sum())} without a listed job); kernel span {end.max() * tick:.1f} us")
loop = (looped - zeroed) * tick
j = joined == 1
print(f"  per joined visit: zero+join {np.mean((zeroed - start)[j] * tick):.2f} us, loop {loop[j].mean():.1f} (max {loop[j].max():.1f}), end {np.mean((end - looped)[j] * tick):.2f} (max {np.max((end - looped)[j] * tick):.2f}) us")
print(f"  chunks walked {int(chunks.sum())}; loop ns per chunk over all visits {1e3 * loop[j].sum() / max(chunks[j].sum(), 1):.1f}")
print(f"  turned back: {int((~j).sum())} visits, {np.sum((end - start)[~j]) * tick:.0f} us-CU")
busy = np.zeros(int(end.max()) + 2, np.int64)
np.add.at(busy, start, 1)
np.add.at(busy, end, -1)
act = np.cumsum(busy)[: int(end.max()) + 1]
T = len(act)
for qd in range(10):
    seg = act[T * qd // 10: T * (qd + 1) // 10]
    print(f"    t {T * qd // 10 * tick:6.1f} .. {T * (qd + 1) // 10 * tick:6.1f} us: workgroups inside a visit, mean {seg.mean():6.1f}")
# per workgroup: when does it leave
last = {}
for i in range(nt):
    last[wg[i]] = max(last.get(wg[i], 0), end[i])
le = np.array(sorted(last.values())) * tick
print(f"  workgroups {len(last)}: last visit ends p10 {np.percentile(le, 10):.1f} p50 {np.percentile(le, 50):.1f} p90 {np.percentile(le, 90):.1f} max {le.max():.1f} us")
hub = rng_id < info.get("hub_ranges", 0)
for name, m in (("hub", hub & j), ("natural", ~hub & j)):
    if m.any():
        print(f"  {name:8s} visits {int(m.sum()):5d} chunks {int(chunks[m].sum()):8d} ns/chunk {1e3 * loop[m].sum() / max(chunks[m].sum(), 1):6.1f}  end phase mean {np.mean((end - looped)[m]) * tick:.2f} us")
print("  last visits to end:")
for t in np.argsort(-end)[:14]:
    print(f"    wg {wg[t]:4d} range {rng_id[t]:4d} {'steal' if stolen[t] else 'list '} {'joined' if joined[t] else 'back  '} chunks {chunks[t]:6d} start {start[t] * tick:6.1f} zero {(zeroed[t] - start[t]) * tick:5.2f} "
          f"loop {loop[t]:6.1f} ({1e3 * loop[t] / max(chunks[t], 1):6.1f} ns/chunk) end-phase {(end[t] - looped[t]) * tick:6.2f} end {end[t] * tick:6.1f}")
print("  first visits of workgroup 0..3:")
for w in range(4):
    for t in np.nonzero(wg == w)[0][np.argsort(start[wg == w])][:6]:
        print(f"    wg {w} range {rng_id[t]:4d} {'steal' if stolen[t] else 'list '} {'joined' if joined[t] else 'back  '} chunks {chunks[t]:6d} start {start[t] * tick:6.1f} loop {loop[t]:6.1f} end {end[t] * tick:6.1f}")
