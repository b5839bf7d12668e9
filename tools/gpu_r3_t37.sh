#!/bin/bash
# round 3: the unsplit form of the sharded product for two ranks: shard tests, the C++ driver, the two-rank run at the driver's N = 2 size and at the small size
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== shard tests"; timeout 1500 python -m pytest tests/test_shard_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
  echo "== two ranks, full size"; bash tools/gpu_r3_t36.sh > /dev/null 2>&1; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_n2_fullsize_gloo_debug.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"], d["roofline"]["kernel"], d["roofline"]["kernel_avg_us"], d["exchange"].get("two_part_form"), {k: d["exchange"].get(k) for k in ("exchange_avg_us", "local_part_avg_us", "remote_part_avg_us", "step_avg_us")})
for v in d.get("variants", []):
    print("  var", v.get("metric"), v.get("value"), v.get("verified"), v["roofline"].get("kernel"), v["roofline"].get("kernel_avg_us"))
PY
  echo "== two ranks, small"; timeout 1200 bash tools/gpu_n2_debug_r03.sh 2>&1 | grep -o '"verified": [a-z]*\|"two_part_form": [a-z]*\|"value": [0-9.]*' | paste - - - - - - | head -4
} 2>&1 | tee gpurun_out/r3_t37.txt
