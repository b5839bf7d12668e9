"""Experiment: share of the gathers v[Aj[k]] that fall on the H hottest columns (in-degree order) of R-MAT graphs."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from spla_amd import graphgen
dev = "cuda" if torch.cuda.is_available() else "cpu"
for scale in [int(a) for a in sys.argv[1:]] or [20]:
    n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device=dev)
    nnz = int(Ap[-1])
    indeg = torch.bincount(Aj.to(torch.int64), minlength=n)
    s, _ = torch.sort(indeg, descending=True)
    c = torch.cumsum(s, 0).to(torch.float64) / nnz
    print(f"scale {scale}: n={n} nnz={nnz} max indeg={int(s[0])}")
    for H in [1 << 10, 1 << 12, 1 << 13, 1 << 14, 24576, 28672, 1 << 15, 40960, 1 << 16, 1 << 17, 1 << 18, 1 << 20]:
        if H <= n:
            print(f"   top {H:8d} columns ({H * 4 // 1024:5d} KiB): {100 * float(c[H - 1]):5.1f}% of entries")
