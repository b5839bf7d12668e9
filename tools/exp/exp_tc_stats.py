import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench_workloads as bw
from spla_amd import graphgen
torch.cuda.set_device(0)
scale = 20
n, Ap, Aj = bw._sym_rmat(scale, 16, 1)
rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (Ap[1:] - Ap[:-1]).to(torch.int64))
low = Aj.to(torch.int64) < rows
key = rows[low] * n + Aj[low].to(torch.int64)
Lp, Lj = graphgen.keys_to_csr(key, n, n)
deg = (Lp[1:] - Lp[:-1]).to(torch.int64)
li = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
la, lb = deg[li], deg[Lj.to(torch.int64)]
heavy = (la + lb > 192) & (la > 0) & (lb > 0)
mn, mx = torch.minimum(la, lb)[heavy], torch.maximum(la, lb)[heavy]
print("nnz(L)", int(Lp[-1]), "max row", int(deg.max()), "heavy entries", int(heavy.sum()))
print("sum min", int(mn.sum()), "sum max", int(mx.sum()), "sum min*log2(max)", float((mn.double() * torch.log2(mx.double())).sum()))
for t in (1000, 5000, 20000):
    print(f"entries with longer row > {t}:", int((mx > t).sum()), " their sum(min):", int(mn[mx > t].sum()))
q = torch.tensor([0.5, 0.9, 0.99, 0.999], device="cuda", dtype=torch.float64)
print("min-len quantiles", torch.quantile(mn.double()[:: max(1, mn.numel() // 1000000)], q).tolist(), "max", int(mn.max()))
# how often is the longer row the A row (i) vs the B row (j)
print("longer is A row:", int((la >= lb)[heavy].sum()), " longer is B row:", int((la < lb)[heavy].sum()))
# size classes of the longer row (LDS staging): entries, sum(longer), sum(shorter), sum(shorter * log2(longer))
for cap in (512, 1024, 2048, 4096, 8192, 16384, 1 << 30):
    m = mx <= cap
    print(f"longer <= {cap}: entries {int(m.sum())}  sum(longer) {int(mx[m].sum())}  sum(shorter) {int(mn[m].sum())}  "
          f"sum(shorter*log2(longer)) {float((mn[m].double() * torch.log2(mx[m].double())).sum()):.3e}")
# row form: heavy entries per mask row, and how long the A row of those rows is
hi = li[heavy]
cnt = torch.bincount(hi, minlength=n)
rows_h = cnt > 0
print("rows with heavy entries", int(rows_h.sum()), " mean heavy entries per such row", float(cnt[rows_h].double().mean()), " max", int(cnt.max()))
print("sum over heavy entries of len(B row)", int(lb[heavy].sum()), " of len(A row)", int(la[heavy].sum()))
for cap in (1024, 4096, 16384):
    m = (deg[hi] <= cap)
    print(f"A row <= {cap}: heavy entries {int(m.sum())}, sum len(B) {int(lb[heavy][m].sum())}, distinct rows {int((rows_h & (deg <= cap)).sum())}, sum len(A) over distinct rows {int(deg[rows_h & (deg <= cap)].sum())}")
