"""CPU simulation, second question: tiles cut by nnz (about E entries each, at most Rmax consecutive rows; a row longer than E is cut by
column) instead of fixed row ranges.  Every tile is then its own row range: no partial slabs, r written once.  How many L1-missing
requests does the column-sorted walk of such tiles need, compared with fixed ranges of R rows (sim_requests.py)?"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from spla_amd import graphgen  # noqa: E402
from tools.exp.sim_requests import distinct_lines_per_wave  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n, Ap, Aj = graphgen.rmat_csr(scale, seed=1)
    Ap = Ap.numpy().astype(np.int64)
    Aj = Aj.numpy().astype(np.int64)
    nnz = Aj.size
    deg = np.diff(Ap)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)
    print(f"scale {scale}: n={n} nnz={nnz}", flush=True)
    for E in (32768, 65536, 131072):
        for Rmax in (16384, 32768):
            t0 = time.time()
            # greedy: consecutive rows while entries <= E and rows <= Rmax
            tile_of_row = np.empty(n, dtype=np.int64)
            t = 0
            cnt = 0
            r0 = 0
            # vectorised greedy via cumulative sums is awkward; loop over rows in numpy chunks instead
            cum = Ap
            i = 0
            starts = [0]
            while i < n:
                # furthest j with cum[j] - cum[i] <= E and j - i <= Rmax
                j = int(np.searchsorted(cum, cum[i] + E, side="right")) - 1
                j = min(j, i + Rmax, n)
                if j <= i:
                    j = i + 1  # a single row longer than E: its own tile(s) (cut by column; counted as one range here)
                tile_of_row[i:j] = t
                t += 1
                i = j
                starts.append(i)
            n_tiles = t
            tl = tile_of_row[rows]
            key = tl * n + Aj
            o = np.argsort(key, kind="stable")
            line = tl[o] * (n // 32 + 1) + (Aj[o] >> 5)
            req = distinct_lines_per_wave(line)
            sizes = np.bincount(tl, minlength=n_tiles)
            print(f"  E={E:7d} Rmax={Rmax:6d}: tiles {n_tiles:6d} (max entries {sizes.max()}, rows written once)  requests {req / 1e6:6.2f} M"
                  f"  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
