"""Round 4: host-side latency of one operator-API call (dispatcher -> format validation -> launch), measured on small inputs where the kernels are negligible:
a dense v_assign_masked (asynchronous), v_count_mf (one synchronisation), a small-frontier vxm (push form: seven launches + one synchronisation)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402,F401
from spla_amd import api  # noqa: E402

n = 1 << 16
rng = np.random.default_rng(1)
v = api.Vector.dense(n, api.INT, 0)
mask = api.Vector.from_lists(np.arange(n), (rng.random(n) < 0.5).astype(np.int32), n, api.INT)
one = api.Scalar(api.INT, 1)
deg = 8
I = np.repeat(np.arange(n, dtype=np.uint32), deg)
J = rng.integers(0, n, n * deg).astype(np.uint32)
A = api.Matrix.from_lists(I, J, np.ones(n * deg, np.int32), (n, n), api.INT)
front = api.Vector.from_lists(np.arange(0, n, 64), np.ones(n // 64, np.int32), n, api.INT)


def timed(fn, reps=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"v_assign_masked (dense, n = {n}; no synchronisation): {timed(lambda: v.assign(mask, one, api.INT.SECOND, api.INT.NQZERO)):7.1f} us per call")
print(f"v_count_mf (dense; one synchronisation):               {timed(lambda: v.count_mf(), 500):7.1f} us per call")
desc = api.Descriptor(early_exit=True, struct_only=True)
out = api.Vector(n, api.INT)
zero = api.Scalar(api.INT, 0)
print(f"vxm_masked, 1024-entry frontier (push; one sync):       {timed(lambda: front.vxm(v, A, api.INT.BAND, api.INT.BOR, api.INT.EQZERO, out=out, init=zero, desc=desc), 500):7.1f} us per call")
t = timed(lambda: api.bfs(A, 0), 200)
print(f"spla_Algorithm_bfs on this graph:                        {t:7.1f} us per run")
