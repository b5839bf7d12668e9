"""Randomised soak of spla_Algorithm_bfs / sssp through the operator API against the oracle: random symmetric graphs (isolated vertices, stars, chains, dense clumps),
every traversal mode, random front factors — the host-decision paths of the traversal loops (polled result words, v_count_mf by compaction, small uploads).
    python tools/exp/fuzz_algos.py [seconds] [seed]"""
import os, sys, time
os.environ["SPLA_ENV_AUTOSYNC"] = "1"
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from oracle import pyoracle
from spla_amd import api, _lib
from test_algorithms import sym_graph
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed)
orc = pyoracle.Oracle()
t0, runs = time.time(), 0
while time.time() - t0 < budget:
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 257, 1000, 1025, 4099, 20000, 70001]))
    deg = int(rng.choice([1, 2, 3, 6, 30]))  # (1: half an edge per vertex — chains and many unreachable vertices)
    Ap, Aj, Ax = sym_graph(rng, n, deg, weights=True) if n > 1 else (np.zeros(2, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    if n > 64 and rng.random() < 0.3:  # a star: one vertex adjacent to a third of the others (a level that reaches most of the graph at once)
        hub = int(rng.integers(0, n))
        leaves = np.setdiff1d(rng.choice(n, n // 3, replace=False), [hub]).astype(np.int64)
        I = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap.astype(np.int64)))
        key = np.unique(np.concatenate([I * n + Aj, hub * n + leaves, leaves * n + hub]))
        I2, J2 = (key // n).astype(np.uint32), (key % n).astype(np.uint32)
        Ap = np.zeros(n + 1, np.uint32); np.cumsum(np.bincount(I2, minlength=n), out=Ap[1:])
        lo, hi = np.minimum(I2, J2).astype(np.int64), np.maximum(I2, J2).astype(np.int64)
        Aj, Ax = J2, (1 + (lo * 31 + hi * 17) % 61).astype(np.float32) / np.float32(4)
    I = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap.astype(np.int64)))
    src = int(rng.integers(0, n))
    ff = float(rng.choice([0.0, 0.01, 0.1, 0.5, 1.0]))
    os.environ["SPLAHIP_MAILBOX_POLL_US"] = str(int(rng.choice([0, 1, 100])))
    os.environ["SPLA_HIP_COB_AFTER"] = str(int(rng.choice([0, 1, 2])))
    if len(Aj):
        A = api.Matrix.from_lists(I, Aj, np.ones(len(Aj), np.int32), (n, n), api.INT)
        W = api.Matrix.from_lists(I, Aj, Ax, (n, n), api.FLOAT)
        want_b, want_s = orc.bfs(n, Ap, Aj, src), orc.sssp(n, Ap, Aj, Ax, src)
        for mode in (api.Descriptor.PUSH, api.Descriptor.PULL, api.Descriptor.PUSH_PULL):
            for rep in range(2):  # the second call runs on the matrix's layout copies
                got = api.bfs(A, src, api.Descriptor(traversal_mode=mode, front_factor=ff)).to_dense()
                assert np.array_equal(got, want_b), ("bfs", n, deg, src, mode, ff, rep, seed, runs)
                got = api.sssp(W, src, api.Descriptor(traversal_mode=mode, front_factor=ff)).to_dense()
                assert np.array_equal(got.view(np.uint32), np.asarray(want_s, np.float32).view(np.uint32)), ("sssp", n, deg, src, mode, ff, rep, seed, runs)
    runs += 1
print(f"fuzz_algos: {runs} graphs x 3 modes x 2 calls x (bfs, sssp) in {time.time() - t0:.0f} s, seed {seed}: all equal to the oracle")
