"""Round 4: where does the first-hit form of a BFS-like step pay?  R-MAT `scale`, structure-only matrix, BAND / BOR int32, mask = EQZERO over a vector that marks
the `visited` share of the rows (the heaviest first, as a traversal finds them), frontier = a random share of the columns / the heaviest columns."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from spla_amd import _lib, graphgen, kernels  # noqa: E402
from tools.exp.exp_r4_cob import timeit  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
_lib.load()
_lib.call("splahip_init", 0)
n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda")
nnz = int(Ap[-1])
ones = torch.ones(nnz, dtype=torch.int32, device="cuda")
deg = (Ap[1:] - Ap[:-1]).to(torch.int64)
order = torch.argsort(deg, descending=True)
gen = torch.Generator(device="cuda")
gen.manual_seed(7)
plan = kernels.CsrPlan(n, nnz, Ap, n, Aj, ones)  # with the sliced-ELL copy (what the BFS-step line attaches)
r = torch.empty(n, dtype=torch.int32, device="cuda")
for visited in (0.02, 0.1):
    depth = torch.zeros(n, dtype=torch.int32, device="cuda")
    depth[order[: int(visited * n)]] = 1
    for kind in ("random", "heaviest"):
        for dens in (0.002, 0.023, 0.1, 0.4):
            k = max(1, int(dens * n))
            front = torch.zeros(n, dtype=torch.int32, device="cuda")
            cols = torch.randperm(n, generator=gen, device="cuda")[:k] if kind == "random" else order[int(visited * n): int(visited * n) + k]
            front[cols] = 1
            out = {}
            for fh, pct in (("0", "15"), ("1", "100"), ("1", "15")):
                os.environ["SPLAHIP_FIRST_HIT"] = fh
                os.environ["SPLAHIP_FIRST_HIT_TAIL_PCT"] = pct
                for _ in range(3):
                    kernels.mxv_masked(plan, "INT", n, Ap, Aj, ones, depth, front, r, "BAND", "BOR", "EQZERO", 0)
                torch.cuda.synchronize()
                us, _ = timeit(lambda: kernels.mxv_masked(plan, "INT", n, Ap, Aj, ones, depth, front, r, "BAND", "BOR", "EQZERO", 0))
                out[fh + pct] = (us, plan.first_hit(), int(r.sum()))
            assert out["015"][2] == out["1100"][2] == out["115"][2]
            fhv = out["1100"][1]
            print(f"visited {visited:4.2f} frontier {kind:8s} {100 * dens:6.2f} % of the columns: sliced-ELL {out['015'][0]:7.1f} us   first hit forced {out['1100'][0]:7.1f} us   "
                  f"default rule {out['115'][0]:7.1f} us (taken {out['115'][1]['taken']}, tail {fhv['tail_pct']} %)   rows reached {out['115'][2]}", flush=True)
plan.close()
