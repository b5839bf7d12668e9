"""CPU simulation behind the row grouping of the column-ordered blocks (round 3; mxv_cob.hpp): how many distinct 128-byte lines of v does a range of rows
touch, summed over the ranges — the number of L1-missing gather requests of one SpMV, whatever the schedule — for candidate groupings of the rows:
fixed ranges of R consecutive rows (round 2), consecutive rows with the empty ones skipped, all rows sorted by degree, and the shipped form: the H rows of
highest degree as degree-sorted "hub" ranges + the other non-empty rows in row order.

    python tools/exp/sim_ranges.py 22        (R-MAT scale; generation on the CPU takes ~30 s at scale 22)

Measured (R-MAT 22, 65.2 M nnz; 131 K lines of v):  fixed 2^14: 6.48 M lines   fixed 2^15: 4.52 M   non-empty rows 2^14: 5.25 M   full degree sort R = 20480: 2.73 M
hub (328 K rows) + non-empty natural R = 20480: 3.00 M   — the shipped grouping (R = 20224, <= 16 hub ranges of rows with degree >= max(16, 2 x average)).
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from spla_amd import graphgen  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n, Ap, Aj = graphgen.rmat_csr(scale, seed=1)
    Ap, Aj = Ap.numpy().astype(np.int64), Aj.numpy().astype(np.int64)
    nnz = Aj.size
    deg = np.diff(Ap)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    nl = (n + 31) // 32
    line = Aj >> 5
    print(f"R-MAT {scale}: n={n} nnz={nnz} empty rows {np.mean(deg == 0):.3f} empty columns {np.mean(np.bincount(Aj, minlength=n) == 0):.3f}")

    def lines(range_of_row, label):
        u = np.unique(range_of_row[rows] * nl + line)
        print(f"  {label}: ranges {int(range_of_row.max()) + 1}  distinct (range, line) pairs {u.size / 1e6:.2f} M = {u.size * 128 / 1e6:.0f} MB", flush=True)

    order = np.argsort(-deg, kind="stable")
    ne = deg > 0
    for R in (16384, 20480):
        lines(np.arange(n) // R, f"fixed, {R} consecutive rows")
        lines(np.maximum(np.cumsum(ne) - 1, 0) // R, f"{R} consecutive non-empty rows")
        r_of = np.empty(n, np.int64)
        r_of[order] = np.arange(n) // R
        lines(r_of, f"all rows sorted by degree, R = {R}")
        for H in (R, 4 * R, 16 * R):
            hub = np.zeros(n, bool)
            hub[order[:H]] = True
            r_of = np.empty(n, np.int64)
            r_of[order[:H]] = np.arange(H) // R
            rest = (~hub) & ne
            r_of[~hub] = H // R + np.maximum(np.cumsum(rest) - 1, 0)[~hub] // R
            lines(r_of, f"{H} hub rows ({deg[hub].sum() / nnz:.3f} of the entries) + non-empty natural rows, R = {R}")


if __name__ == "__main__":
    main()
