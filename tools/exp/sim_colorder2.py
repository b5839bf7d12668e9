"""Round 4 sim: distinct (range, line) pairs when only the HOT columns (in-degree >= thr) are relabelled by descending in-degree into a compact array
and the other columns keep their labels (gathered from the caller's v), for raw and random labels.   python tools/exp/sim_colorder2.py 21"""
import sys

import numpy as np

sys.path.insert(0, ".")
from spla_amd import graphgen  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    R, Rn = 20224, 32704
    n, Ap, Aj = graphgen.rmat_csr(scale, seed=1)
    Ap, Aj = Ap.numpy().astype(np.int64), Aj.numpy().astype(np.int64)
    nnz = Aj.size
    deg = np.diff(Ap)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    nl = (n + 31) // 32
    thr = max(16, 2 * nnz // n)
    order = np.argsort(-deg, kind="stable")
    H = min(int((deg >= thr).sum()), 32 * R)
    hub = np.zeros(n, bool)
    hub[order[:H]] = True
    ne = deg > 0
    r_of = np.empty(n, np.int64)
    r_of[order[:H]] = np.arange(H) // R
    nh = (H + R - 1) // R
    rest = (~hub) & ne
    r_of[~hub] = nh + np.maximum(np.cumsum(rest) - 1, 0)[~hub] // Rn
    rr = r_of[rows]
    is_nat = rr >= nh
    print(f"R-MAT {scale}: n={n} nnz={nnz} hub rows {H} in {nh} ranges, natural ranges {int(r_of.max()) + 1 - nh}")
    rng = np.random.default_rng(3)
    for lname, lab in (("raw", np.arange(n)), ("random", rng.permutation(n))):
        col = lab[Aj]
        indeg = np.bincount(col, minlength=n)
        byp = np.argsort(-indeg, kind="stable")
        for cthr in (0, 64, 16, 8, 4, 2, 1):
            if cthr == 0:
                K = 0
                line = col >> 5
            else:
                K = int((indeg >= cthr).sum())
                rank = np.empty(n, np.int64)
                rank[byp] = np.arange(n)
                hot = rank[col] < K
                line = np.where(hot, rank[col] >> 5, nl + (col >> 5))
            key = rr * (2 * nl) + line
            u_all = np.unique(key).size
            u_nat = np.unique(key[is_nat]).size
            cov = float((indeg[col] >= cthr).mean()) if cthr else 0.0
            print(f"  {lname:6s} labels, hot = in-degree >= {cthr:3d}: K = {K / 1e6:5.2f} M columns ({cov:.3f} of entries)  pairs {u_all / 1e6:6.2f} M (natural {u_nat / 1e6:6.2f} M)"
                  f"   pairs + K = {(u_all + K) / 1e6:6.2f} M", flush=True)


if __name__ == "__main__":
    main()
