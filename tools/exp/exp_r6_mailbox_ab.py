"""Round 6: spla_Algorithm_bfs / sssp end to end with the mailbox words polled (SPLAHIP_MAILBOX_POLL_US=100) against a stream synchronisation (0), alternating IN ONE
PROCESS (the box's clocks drift over the first seconds: separate processes confound the comparison)."""
import os, sys, time
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
A = api.Matrix.from_lists(I_h, Aj_h, np.ones(nnz, np.int32), (n, n), api.INT)
Af = api.Matrix.from_lists(I_h, Aj_h, (1.0 + (np.arange(nnz) % 7)).astype(np.float32), (n, n), api.FLOAT)
src = 0
def run(f, reps):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for _ in range(30): api.bfs(A, src)
for rnd in range(5):
    for us in ("0", "100", "400"):
        os.environ["SPLAHIP_MAILBOX_POLL_US"] = us; _lib.env_refresh()
        print(f"round {rnd} poll_us {us:>3}: bfs {run(lambda: api.bfs(A, src), 100):.4f} ms   sssp {run(lambda: api.sssp(Af, src), 20):.4f} ms", flush=True)
