"""Where does an end-to-end BFS level spend its time?  Manual bfs loop through the operator API with a device sync after every op."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
import bench_workloads as bw
from spla_amd import api, _lib
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, Ap, Aj = bw._sym_rmat(scale, 16, 1)
nnz = int(Ap[-1])
Ap_h, Aj_h = Ap.cpu().numpy().astype(np.uint32), Aj.cpu().numpy().astype(np.uint32)
I_h = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap_h.astype(np.int64)))
A = api.Matrix.from_lists(I_h, Aj_h, np.ones(nnz, np.int32), (n, n), api.INT)
src = int(np.argmax(np.diff(Ap_h.astype(np.int64))))
desc = api.Descriptor(early_exit=True, struct_only=True)

def tick():
    torch.cuda.synchronize(); return time.perf_counter()

for rep in range(3):
    depth = api.Vector(n, api.INT); front = api.Vector(n, api.INT); front.set(src, 1)
    size, level, rows = 1, 1, []
    t_all = tick()
    while True:
        t0 = tick()
        depth.assign(front, level, api.INT.SECOND, api.INT.NQZERO)
        t1 = tick()
        push = size / n <= 0.1
        if push:
            new = front.vxm(depth, A, api.INT.BAND, api.INT.BOR, api.INT.EQZERO, desc=desc)
        else:
            new = A.mxv(depth, front, api.INT.BAND, api.INT.BOR, api.INT.EQZERO, desc=desc)
        t2 = tick()
        size = new.count_mf()
        t3 = tick()
        rows.append((level, "push" if push else "pull", size, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
        if size == 0: break
        front, level = new, level + 1
    total = (tick() - t_all) * 1e6
    if rep == 2:
        for r in rows: print("level %d %s new=%d  assign %.0f us  product %.0f us  count %.0f us" % r)
        print("total %.0f us" % total)
