"""Host threads: the library keeps its per-call workspaces per thread (scratch arena of mxm / mxmT, pinned staging arena, cache of released stream-ordered
blocks — runtime.hip, mxm_spa.hip), so calls of different host threads on streams of their own must not disturb one another."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(scale, seed):
    from spla_amd import graphgen
    n, Ap, Aj = graphgen.rmat_csr(scale, 8, seed=seed, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    Ax = torch.randint(1, 5, (int(Ap[-1]),), generator=g, device="cuda", dtype=torch.int32)
    return n, (Ap, Aj, Ax)


def _work(scale, seed):
    """mxm (row classes: pinned arena + scratch arena), mxmT (scratch arena), a push vxm (stream-ordered blocks) and an mxv of one small R-MAT matrix."""
    from spla_amd import kernels
    n, A = _inputs(scale, seed)
    R, _ = kernels.mxm("INT", n, n, A, A, "MULT", "PLUS", 0)
    T = kernels.mxmT_masked("INT", n, A, A, A, "MULT", "PLUS", "ALWAYS", 0)
    v = torch.ones(n, dtype=torch.int32, device="cuda")
    r = torch.empty(n, dtype=torch.int32, device="cuda")
    plan = kernels.CsrPlan(n, int(A[0][-1]), A[0])
    kernels.mxv_masked(plan, "INT", n, A[0], A[1], A[2], torch.zeros(n, dtype=torch.int32, device="cuda"), v, r, "MULT", "PLUS", "ALWAYS", 0, False)
    vi = torch.arange(0, n, 97, dtype=torch.int32, device="cuda")  # a small frontier: the single-pass push form (its per-thread all-zero accumulators)
    P = kernels.vxm_push_small("INT", n, n, A[0], A[1], A[2], torch.zeros(n, dtype=torch.int32, device="cuda"), vi, torch.ones_like(vi), "MULT", "MAX", "ALWAYS", 1 << 22)
    torch.cuda.current_stream().synchronize()
    plan.close()
    assert P[0], "the small-frontier push form was expected to run"
    return [t.cpu().numpy() for t in (*R, *T, r, P[3], P[4])]


def test_calls_of_four_host_threads_on_their_own_streams_match_the_single_threaded_results():
    keys = ((13, 1), (12, 2))
    want = {k: _work(*k) for k in keys}
    errors = []

    def runner(key):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(6):
                    for a, b in zip(_work(*key), want[key]):
                        if a.shape != b.shape or not np.array_equal(a, b):
                            errors.append(f"{key}: a result differs from the single-threaded one")
                            return
        except Exception as e:  # noqa: BLE001  (reported by the assert below)
            errors.append(f"{key}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=runner, args=(k,)) for k in keys + keys]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:4]


def test_caches_of_a_thread_that_has_exited_are_given_back_by_trim():
    """the per-thread caches (scratch arena, released stream-ordered blocks, pinned staging chunks) are listed process-wide; a host thread that did mxm /
    mxmT / push work and then exited leaves its caches as orphans, and splahip_mxm_trim — called from ANOTHER thread — gives their device memory back to the driver
    (hipMemGetInfo recovers), as an out-of-memory retry of any thread would.  Before: thread_locals created with `new`, trimmed only by their own thread, never freed."""
    from spla_amd import _lib
    _work(13, 3)  # (warm: kernels loaded, this thread's own caches in steady state)
    _lib.call("splahip_mxm_trim")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    held = []

    def runner():
        with torch.cuda.stream(torch.cuda.Stream()):
            _work(14, 4)
            _work(14, 4)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        held.append(free0 - torch.cuda.mem_get_info()[0])

    t = threading.Thread(target=runner)
    t.start()
    t.join()
    assert held and held[0] > (8 << 20), f"the worker thread was expected to leave cached device memory behind (held {held})"
    _lib.call("splahip_mxm_trim")  # from the main thread: the exited thread's caches are orphans in the registry
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (16 << 20), f"device memory not recovered after the trim: {held[0] >> 20} MiB held by the exited thread, {(free0 - free1) >> 20} MiB still missing"
    # and the library keeps working on both kinds of thread afterwards
    a = _work(12, 2)
    out = []
    t2 = threading.Thread(target=lambda: out.append(_work(12, 2)))
    t2.start()
    t2.join()
    assert all(np.array_equal(x, y) for x, y in zip(a, out[0]))
