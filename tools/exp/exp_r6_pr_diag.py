import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import bench_workloads as bw
from spla_amd import api, _lib
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
n, Ap, Aj = bw._sym_rmat(20, 16, 1)
nnz = int(Ap[-1])
Ap_h, Aj_h = Ap.cpu().numpy().astype(np.uint32), Aj.cpu().numpy().astype(np.uint32)
I_h = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap_h.astype(np.int64)))
deg = np.maximum(np.diff(Ap_h.astype(np.int64)), 1).astype(np.float32)
A = api.Matrix.from_lists(I_h, Aj_h, (0.85 / deg[Aj_h]).astype(np.float32), (n, n), api.FLOAT)
for sp in ("1", "0", "1", "0"):
    os.environ["SPLAHIP_SCAN_POST"] = sp; _lib.env_refresh()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = api.pr(A, 0.85, 1e-5)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        d = p.to_dense()
        print(f"scan_post {sp} rep {rep}: {dt*1e3:.3f} ms  sum {float(d.sum()):.6f} max {float(d.max()):.3e}", flush=True)
