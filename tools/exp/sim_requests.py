"""CPU simulation: how many L1-missing 128-byte requests does a row-blocked, rank-sorted SpMV layout need?

For R-MAT scale S: rows are cut into ranges of R rows; a range's entries are sorted by column RANK (descending reference count),
ranks < H are served from LDS (no request); for the rest, every aligned group of 64 consecutive entries (one wave load) costs one
request per distinct 128-byte line of the rank-ordered vector.  Prints request counts for several (R, H) and for natural column
order, so the layout can be chosen before a kernel is written.
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from spla_amd import graphgen  # noqa: E402


def distinct_lines_per_wave(keys_sorted_line, group=64):
    """keys_sorted_line: int64 array of line ids in processing order; count distinct lines per aligned group of `group`."""
    n = keys_sorted_line.size
    pad = (-n) % group
    if pad:
        keys_sorted_line = np.concatenate([keys_sorted_line, np.full(pad, -1, dtype=keys_sorted_line.dtype)])
    a = keys_sorted_line.reshape(-1, group)
    a = np.sort(a, axis=1)
    d = (a[:, 1:] != a[:, :-1]).sum(axis=1) + 1
    return int(d.sum())


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    t0 = time.time()
    n, Ap, Aj = graphgen.rmat_csr(scale, seed=1)
    Ap = Ap.numpy().astype(np.int64)
    Aj = Aj.numpy().astype(np.int64)
    nnz = Aj.size
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
    print(f"scale {scale}: n={n} nnz={nnz} gen {time.time() - t0:.1f}s", flush=True)
    cnt = np.bincount(Aj, minlength=n)
    order = np.argsort(-cnt, kind="stable")  # rank -> column
    rank_of = np.empty(n, dtype=np.int64)
    rank_of[order] = np.arange(n)
    rk = rank_of[Aj]
    csum = np.cumsum(cnt[order])
    for h in (16384, 24576, 32768, 40000, 65536, 1 << 20):
        print(f"  top {h} ranks cover {csum[h - 1] / nnz:.3f} of entries")
    stream_lines = nnz * 8 // 128
    print(f"  entry stream lines (8 B/entry): {stream_lines / 1e6:.2f} M")
    for R in (8192, 16384, 32768, 40960):
        rng = rows // R
        for label, col in (("rank", rk), ("natural", Aj)):
            key = rng * n + col
            o = np.argsort(key, kind="stable")
            cs = col[o]
            rs = rng[o]
            for H in ((0, 24576, 40000) if label == "rank" else (0,)):
                m = cs >= H
                # a wave group must not straddle ranges in this estimate: key the line by (range, line)
                line = rs[m] * (n // 32 + 1) + (cs[m] >> 5)
                req = distinct_lines_per_wave(line)
                # lower bound if requests merged across waves too (distinct (range, line) pairs)
                uniq = int((np.diff(line) != 0).sum() + 1)
                print(f"  R={R:6d} {label:8s} H={H:6d}: cold entries {m.sum() / 1e6:6.2f} M  requests/wave-merged {req / 1e6:6.2f} M"
                      f"  distinct (range,line) {uniq / 1e6:6.2f} M", flush=True)


if __name__ == "__main__":
    main()
