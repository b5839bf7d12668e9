"""spla_Algorithm_sssp / pr end to end, repeated (for rocprofv3 --hip-trace --kernel-trace --stats)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
import bench_workloads as bw
from spla_amd import api, _lib
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
n, Ap, Aj = bw._sym_rmat(20, 16, 1)
nnz = int(Ap[-1])
Ap_h, Aj_h = Ap.cpu().numpy().astype(np.uint32), Aj.cpu().numpy().astype(np.uint32)
I_h = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap_h.astype(np.int64)))
lo, hi = np.minimum(I_h, Aj_h).astype(np.int64), np.maximum(I_h, Aj_h).astype(np.int64)
W_h = (((lo * 2654435761 + hi * 40503) % 255) + 1).astype(np.float32)
A = api.Matrix.from_lists(I_h, Aj_h, W_h, (n, n), api.FLOAT)
src = int(np.argmax(np.diff(Ap_h.astype(np.int64))))
api.sssp(A, src); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): api.sssp(A, src)
torch.cuda.synchronize()
print("sssp e2e ms", (time.perf_counter() - t0) / 10 * 1e3)
