"""Round 5: the intermittent mismatch of the run-time schedule on the small geometry (tests: cob-small-dyn), reproduced in a loop: which rows differ and how."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from gpu_utils import dev, host
from spla_amd import _lib, graphgen, kernels
os.environ["SPLA_ENV_AUTOSYNC"] = "1"
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
bits = sys.argv[1] if len(sys.argv) > 1 else "2"
workers = sys.argv[2] if len(sys.argv) > 2 else "3"
n, Ap_t, Aj_t = graphgen.rmat_csr(15, 16, seed=1, device="cpu")
Ap, Aj = Ap_t.numpy().astype(np.uint32), Aj_t.numpy().astype(np.uint32)
rng = np.random.default_rng(4)
Ax = rng.integers(1, 256, len(Aj)).astype(np.int32)
INF = 1 << 30
v = np.full(n, INF, np.int32); k = n // 8
v[rng.permutation(n)[:k]] = rng.integers(0, 5000, k).astype(np.int32)
bounds = graphgen.nnz_balanced_row_splits(Ap_t, 2)
bad_runs = 0
for it in range(40):
    for p in range(2):
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        s, e = int(Ap[lo]), int(Ap[hi])
        Ap_b = (Ap[lo:hi + 1].astype(np.int64) - s).astype(np.uint32)
        dAp, dAj, dAx, dm, dv = dev(Ap_b), dev(Aj[s:e].copy()), dev(Ax[s:e].copy()), dev(np.zeros(hi - lo, np.int32)), dev(v)
        outs = {}
        for form in ("csr", "cob-small-dyn"):
            os.environ["SPLAHIP_COB_BITS"] = bits if form != "csr" else "1"
            plan = kernels.CsrPlan(hi - lo, e - s, dAp)
            if form != "csr":
                os.environ["SPLAHIP_COB_DYN"] = "1"; os.environ["SPLAHIP_COB_DYN_WORKERS"] = workers
                os.environ["SPLAHIP_COB_HUB_MIN_DEGREE"] = "3"; os.environ["SPLAHIP_COB_NAT_SLOTS"] = "512"
                plan.attach_cob(n, dAp, dAj, dAx, row_bits=8, tile_chunks=2)
                for kk in ("SPLAHIP_COB_DYN", "SPLAHIP_COB_DYN_WORKERS", "SPLAHIP_COB_HUB_MIN_DEGREE", "SPLAHIP_COB_NAT_SLOTS"):
                    os.environ.pop(kk)
            r = torch.full((hi - lo,), 3, dtype=torch.int32, device="cuda")
            kernels.mxv_masked(plan, "INT", n, dAp, dAj, dAx, dm, dv, r, "PLUS", "MIN", "ALWAYS", INF, False)
            torch.cuda.synchronize()
            outs[form] = host(r, np.int32)
            plan.close()
        d = np.nonzero(outs["csr"] != outs["cob-small-dyn"])[0]
        if len(d):
            bad_runs += 1
            deg = (Ap_b[1:] - Ap_b[:-1])[d]
            print(f"it {it} part {p}: {len(d)} rows differ; rows {d[:8]} deg {deg[:8]} want {outs['csr'][d[:8]]} got {outs['cob-small-dyn'][d[:8]]}; span of rows {d.min()}..{d.max()}", flush=True)
print(f"bits={bits} workers={workers}: {bad_runs} of 80 part-runs differ")
