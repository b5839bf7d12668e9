"""CPU simulation: same-row multiplicity inside one wave instruction of the column-ordered-blocks walk (64 lanes, slots j, j+4, ... of a
256-entry chunk): how many compare-and-swap retries would an LDS float accumulation need?  R-MAT, ranges of R rows, entries by (range, col, row)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from spla_amd import graphgen  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    n, Ap, Aj = graphgen.rmat_csr(scale, seed=1)
    Ap = Ap.numpy().astype(np.int64)
    Aj = Aj.numpy().astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
    key = (rows // R) * n + Aj
    o = np.argsort(key, kind="stable")
    rs = rows[o]
    rng = rs // R
    m = rs.size // 256 * 256
    # (ignoring range / window padding: chunks of 256 consecutive sorted entries; instruction j takes slots j, j+4, ...)
    a = rs[:m].reshape(-1, 64, 4)
    tot_extra = 0
    tot_rounds = 0
    hist = np.zeros(65, np.int64)
    for j in range(4):
        g = np.sort(a[:, :, j], axis=1)
        same = (g[:, 1:] == g[:, :-1])
        # multiplicity of the most frequent row in each instruction
        run = np.zeros(g.shape[0], np.int64)
        cur = np.zeros(g.shape[0], np.int64)
        for k in range(same.shape[1]):
            cur = np.where(same[:, k], cur + 1, 0)
            run = np.maximum(run, cur)
        tot_extra += int(same.sum())
        tot_rounds += int(run.sum())
        hist += np.bincount(run + 1, minlength=65)[:65]
    n_instr = a.shape[0] * 4
    print(f"scale {scale} R={R}: wave instructions {n_instr / 1e6:.2f} M; lanes that lose a CAS round (sum of multiplicity-1) {tot_extra / 1e6:.2f} M "
          f"({tot_extra / rs.size:.3f} per entry); extra serial rounds (max multiplicity - 1 per instruction) {tot_rounds / 1e6:.2f} M "
          f"({tot_rounds / n_instr:.3f} per instruction)")
    print("  max multiplicity histogram (1..10, >10):", hist[1:11].tolist(), int(hist[11:].sum()))


if __name__ == "__main__":
    main()
