#!/bin/bash
# Round 6: order-preserving compaction in one launch (compact_fused.hpp) at the three sites of a traversal level — guarded first run, the tests, then A/B end to end
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_fused.txt
: > $out
timeout 300 python3 -m pytest tests/test_algorithms.py -x -q -m gpu --timeout 200 -k "host_round_trips" 2>&1 | tail -3 | tee -a $out || exit 1
timeout 1500 python3 -m pytest tests/test_algorithms.py tests/test_api_gpu.py tests/test_vxm_gpu.py tests/test_reference_suite_gpu.py tests/test_threads_gpu.py tests/test_user_ops_gpu.py tests/test_golden_next.py -x -q -m gpu --timeout 600 2>&1 | tail -3 | tee -a $out
cat > /tmp/ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import bench_workloads as bw
from spla_amd import api, _lib
torch.cuda.set_device(0); _lib.load(); _lib.call("splahip_init", 0)
n, Ap, Aj = bw._sym_rmat(20, 16, 1)
nnz = int(Ap[-1])
Ap_h, Aj_h = Ap.cpu().numpy().astype(np.uint32), Aj.cpu().numpy().astype(np.uint32)
I_h = np.repeat(np.arange(n, dtype=np.uint32), np.diff(Ap_h.astype(np.int64)))
A = api.Matrix.from_lists(I_h, Aj_h, np.ones(nnz, np.int32), (n, n), api.INT)
Af = api.Matrix.from_lists(I_h, Aj_h, (1.0 + (np.arange(nnz) % 7)).astype(np.float32), (n, n), api.FLOAT)
def run(f, reps):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for _ in range(30): api.bfs(A, 0)
for rnd in range(4):
    for fused in ("0", "1"):
        os.environ["SPLAHIP_FUSED_COMPACT"] = fused; _lib.env_refresh()
        print(f"round {rnd} fused {fused}: bfs {run(lambda: api.bfs(A, 0), 100):.4f} ms   sssp {run(lambda: api.sssp(Af, 0), 20):.4f} ms", flush=True)
PY
timeout 600 python3 /tmp/ab.py 2>&1 | grep round | tee -a $out
for cfg in 1 2 0 1; do
  echo "== SPLA_HIP_COUNT_BY_COMPACTION=$cfg" | tee -a $out
  SPLA_HIP_COUNT_BY_COMPACTION=$cfg timeout 600 python3 bench.py --workload algorithms --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v.get('ms') for k,v in d['algorithms'].items()}, d.get('verified'))" | tee -a $out
done
