"""Round 6: dense -> sparse conversion of a 2^20-entry vector, one launch (compact_fused.hpp) against count / scan / write / read-back (the SPLAHIP_FUSED_COMPACT knob exists only with profiles/r06_fused_compaction_experiment.patch applied); run under rocprofv3 --kernel-trace --stats
for the kernels' own durations."""
import ctypes as C, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from spla_amd import _lib
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
n = 1 << 20
for dens in (0.001, 0.1, 0.5):
    x = (torch.rand(n, device="cuda") < dens).to(torch.int32)
    di = torch.empty(n, dtype=torch.int32, device="cuda"); dv = torch.empty(n, dtype=torch.int32, device="cuda")
    cnt = C.c_uint32()
    for fused in ("0", "1", "0", "1"):
        os.environ["SPLAHIP_FUSED_COMPACT"] = fused; _lib.env_refresh()
        for _ in range(20): _lib.call("splahip_vec_dense_to_coo_limited", 0, n, x.data_ptr(), 0, di.data_ptr(), dv.data_ptr(), n, C.byref(cnt), None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): _lib.call("splahip_vec_dense_to_coo_limited", 0, n, x.data_ptr(), 0, di.data_ptr(), dv.data_ptr(), n, C.byref(cnt), None)
        torch.cuda.synchronize()
        print(f"density {dens}: fused {fused}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call, count {cnt.value} (expected {int(x.sum())})", flush=True)
