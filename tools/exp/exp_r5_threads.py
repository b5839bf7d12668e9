"""Round 5: two host threads, each with a stream of its own, run mxm / mxmT / vxm-push / mxv calls at the same time (ctypes drops the GIL inside a call);
every result must equal the one the same call gives alone.  Exercises the per-thread scratch arena, pinned arena and stream-block cache."""
import sys, threading
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from spla_amd import _lib, graphgen, kernels
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)

def inputs(scale, seed):
    n, Ap, Aj = graphgen.rmat_csr(scale, 8, seed=seed, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    Ax = torch.randint(1, 5, (int(Ap[-1]),), generator=g, device="cuda", dtype=torch.int32)
    return n, (Ap, Aj, Ax)

def work(scale, seed):
    n, A = inputs(scale, seed)
    R, _ = kernels.mxm("INT", n, n, A, A, "MULT", "PLUS", 0)
    T = kernels.mxmT_masked("INT", n, A, A, A, "MULT", "PLUS", "ALWAYS", 0)
    v = torch.ones(n, dtype=torch.int32, device="cuda"); r = torch.empty(n, dtype=torch.int32, device="cuda")
    p = kernels.CsrPlan(n, int(A[0][-1]), A[0])
    kernels.mxv_masked(p, "INT", n, A[0], A[1], A[2], torch.zeros(n, dtype=torch.int32, device="cuda"), v, r, "MULT", "PLUS", "ALWAYS", 0, False)
    torch.cuda.current_stream().synchronize(); p.close()
    return [t.cpu().numpy() for t in (*R, *T, r)]

want = {k: work(*k) for k in ((13, 1), (12, 2))}
errors = []
def runner(key, reps):
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(reps):
                got = work(*key)
                for a, b in zip(got, want[key]):
                    if a.shape != b.shape or not np.array_equal(a, b):
                        errors.append(f"{key}: mismatch"); return
    except Exception as e:  # noqa: BLE001
        errors.append(f"{key}: {type(e).__name__}: {e}")
ts = [threading.Thread(target=runner, args=(k, 12)) for k in ((13, 1), (12, 2), (13, 1), (12, 2))]
[t.start() for t in ts]; [t.join() for t in ts]
print("threads:", "OK" if not errors else errors[:4])
