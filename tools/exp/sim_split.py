"""CPU simulation: long rows (degree >= D) walked in CSR order by whole waves (register accumulation), the rest as column-ordered blocks.
Requests of both parts (distinct 128-B lines of v per 64 consecutive entries) and the same-row multiplicity left in the blocks."""
import sys

import numpy as np

sys.path.insert(0, ".")
from spla_amd import graphgen  # noqa: E402
from tools.exp.sim_requests import distinct_lines_per_wave  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    R = 16384
    n, Ap, Aj = graphgen.rmat_csr(scale, seed=1)
    Ap = Ap.numpy().astype(np.int64)
    Aj = Aj.numpy().astype(np.int64)
    deg = np.diff(Ap)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    nnz = Aj.size
    for D in (128, 512, 2048, 1 << 30):
        long_row = deg >= D
        is_long = long_row[rows]
        # long part: CSR order, groups of 64 consecutive entries of one row (pad rows to 64)
        lr = rows[is_long]
        lc = Aj[is_long]
        req_long = 0
        if lr.size:
            # position inside the row
            first = Ap[lr]
            pos = np.arange(nnz)[is_long] - first
            grp = lr * (1 << 20) + pos // 64
            line = lc >> 5
            k = grp * (n // 32 + 1) + line
            req_long = int(np.unique(k).size)
        sr = rows[~is_long]
        sc = Aj[~is_long]
        key = (sr // R) * n + sc
        o = np.argsort(key, kind="stable")
        line = (sr[o] // R) * (n // 32 + 1) + (sc[o] >> 5)
        req_short = distinct_lines_per_wave(line)
        rs = sr[o]
        m = rs.size // 64 * 64
        g = np.sort(rs[:m].reshape(-1, 64), axis=1)
        same = g[:, 1:] == g[:, :-1]
        coll = same.any(axis=1).mean()
        print(f"scale {scale} D={D:10d}: long rows {int(long_row.sum()):8d} with {lr.size / nnz:.3f} of the entries: {req_long / 1e6:6.2f} M requests; "
              f"blocks: {req_short / 1e6:6.2f} M requests, instructions with a repeated row {coll:.3f}, lanes losing a round {same.sum() / max(1, rs.size):.4f}/entry",
              flush=True)


if __name__ == "__main__":
    main()
