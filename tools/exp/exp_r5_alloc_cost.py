"""Round 5: what a stream-ordered allocation / release costs the host (splahip_malloc = hipMallocAsync, splahip_free = hipFreeAsync), steady state."""
import ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from spla_amd import _lib
torch.cuda.set_device(0); lib = _lib.load(); _lib.call("splahip_init", 0)
for mb in (1, 16, 64, 256):
    for rep in range(3):
        ptrs = [C.c_void_p() for _ in range(20)]
        t0 = time.perf_counter()
        for p in ptrs:
            _lib.call("splahip_malloc", C.byref(p), mb << 20)
        t1 = time.perf_counter()
        for p in ptrs:
            _lib.call("splahip_free", p)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print(f"{mb:4d} MB x 20: alloc {1e6 * (t1 - t0) / 20:7.1f} us each, free {1e6 * (t2 - t1) / 20:7.1f} us each, sync after {1e6 * (t3 - t2):7.1f} us", flush=True)
