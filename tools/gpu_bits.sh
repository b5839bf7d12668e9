#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mxv_gpu.py -m gpu -x -q 2>&1 | tail -12
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_bits.json 2> gpurun_out/bench_bits.err; echo "bench exit $?"; grep -v amdgpu.ids gpurun_out/bench_bits.err | tail -12; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_bits.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"])
for v in d.get("variants",[]): print(v["name"], v["kernel"], v["kernel_avg_us"], v.get("gteps_local"))
PY
