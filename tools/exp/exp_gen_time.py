import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from spla_amd import graphgen
torch.cuda.set_device(0)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for scale, permute in ((19, True), (19, True), (22, False), (22, True), (24, True)):
    t0 = time.time()
    n, Ap, Aj = graphgen.rmat_csr(scale, 16, seed=1, device="cuda", permute=permute)
    torch.cuda.synchronize()
    print(scale, permute, f"{time.time() - t0:.2f}s nnz={int(Ap[-1])}", flush=True)
    del Ap, Aj
t0 = time.time(); p = torch.randperm(1 << 25, device="cuda"); torch.cuda.synchronize(); print("randperm 2^25", time.time() - t0)
