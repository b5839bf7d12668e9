"""CPU simulation (round 4): distinct (range, 128-byte line of v) pairs of the shipped row grouping of the column-ordered blocks under three column
labelings of the SAME graph: the generator's raw labels, random labels (a real graph's), and columns relabelled by descending in-degree (popularity).
    python tools/exp/sim_colorder.py 22
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from spla_amd import graphgen  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    R, Rn = 20224, int(sys.argv[2]) if len(sys.argv) > 2 else 32704
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
    print(f"R-MAT {scale}: n={n} nnz={nnz} hub rows {H} in {nh} ranges ({deg[hub].sum() / nnz:.3f} of entries), natural ranges {int(r_of.max()) + 1 - nh} of {Rn} slots")
    indeg = np.bincount(Aj, minlength=n)
    rng = np.random.default_rng(3)
    perms = {"raw labels": np.arange(n), "random labels": rng.permutation(n), "by descending in-degree": np.argsort(np.argsort(-indeg, kind="stable"))}
    for name, newid in perms.items():
        line = newid[Aj] >> 5
        key = rr * nl + line
        u_all = np.unique(key).size
        u_nat = np.unique(key[is_nat]).size
        print(f"  {name:26s}: pairs {u_all / 1e6:6.2f} M = {u_all * 128 / 1e6:6.0f} MB   (natural {u_nat / 1e6:6.2f} M, hub {(u_all - u_nat) / 1e6:6.2f} M)", flush=True)


if __name__ == "__main__":
    main()
