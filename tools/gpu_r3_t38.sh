#!/bin/bash
# round 3: shipped dist defaults once more: shard tests, C++ drivers, the two-rank run at the driver's N = 2 size (two-part form)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== shard + mxv tests"; timeout 1500 python -m pytest tests/test_shard_gpu.py tests/test_mxv_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
  echo "== two ranks, full size"; bash tools/gpu_r3_t36.sh > /dev/null 2>&1; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_n2_fullsize_gloo_debug.json").read().strip().splitlines()[-1])
print(d["verified"], d["roofline"]["kernel"], d["roofline"]["kernel_avg_us"], d["exchange"].get("two_part_form"), d["exchange"].get("local_part_avg_us"), d["exchange"].get("remote_part_avg_us"))
for v in d.get("variants", []):
    print("  var", v["verified"], v["roofline"]["kernel"], v["roofline"]["kernel_avg_us"])
PY
} 2>&1 | tee gpurun_out/r3_t38.txt
