"""MatrixMarket loader: this library's MtxLoader against the reference's on the same files — comments,
a value column, self loops, duplicates, every option combination; bit-identical edge lists expected."""
import itertools

import numpy as np
import pytest


def _write(path, n, entries, with_values):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% a comment line\n%another\n")
        f.write(f"{n} {n} {len(entries)}\n")
        for k, (i, j) in enumerate(entries):
            f.write(f"{i} {j} {0.5 + k}\n" if with_values else f"{i} {j}\n")


@pytest.mark.parametrize("with_values", [False, True])
def test_loader_equals_reference(reference, tmp_path, with_values):
    from spla_amd import api
    rng = np.random.default_rng(3)
    n = 300
    entries = [(int(a) + 1, int(b) + 1) for a, b in zip(rng.integers(0, n, 2500), rng.integers(0, n, 2500))]
    entries += [(7, 7), (8, 8), entries[0], entries[1]]  # self loops and duplicates
    path = tmp_path / "m.mtx"
    _write(path, n, entries, with_values)
    for offset, undirected, loops in itertools.product([True, False], repeat=3):
        got = api.load_mtx(path, offset, undirected, loops)
        want = reference.load_mtx(path, offset, undirected, loops)
        assert got[:2] == want[:2]
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), (offset, undirected, loops)
    nr, nc, Ai, Aj = api.load_mtx(path)
    key = Ai.astype(np.int64) * n + Aj
    assert np.all(np.diff(key) > 0) and not np.any(Ai == Aj)  # sorted, unique, loop-free
    assert set(zip(Ai.tolist(), Aj.tolist())) == {(b, a) for a, b in zip(Ai.tolist(), Aj.tolist())}  # symmetric
