"""Round 6: spla_Algorithm_sssp / pr end to end, a few runs each (for rocprofv3 --kernel-trace: the timeline of one run is printed by tools/gpu_r6_algo_trace.sh)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
import bench_workloads as bw
from spla_amd import api, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "sssp"
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
n, Ap, Aj = bw._sym_rmat(20, 16, 1)
nnz = int(Ap[-1])
Ap_h, Aj_h = Ap.cpu().numpy().astype(np.uint32), Aj.cpu().numpy().astype(np.uint32)
I_h = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap_h.astype(np.int64)))
rng = np.random.default_rng(3)
if which == "sssp":
    A = api.Matrix.from_lists(I_h, Aj_h, rng.integers(1, 16, nnz).astype(np.float32), (n, n), api.FLOAT)
    f = lambda: api.sssp(A, 0)
else:
    deg = np.maximum(np.diff(Ap_h.astype(np.int64)), 1).astype(np.float32)
    A = api.Matrix.from_lists(I_h, Aj_h, (0.85 / deg[Aj_h]).astype(np.float32), (n, n), api.FLOAT)
    f = lambda: api.pr(A, 0.85, 1e-5)
f(); f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6): f()
torch.cuda.synchronize()
print(which, "e2e ms", (time.perf_counter() - t0) / 6 * 1e3)
