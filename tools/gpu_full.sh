#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_bits.json 2> gpurun_out/bench_bits.err; echo "bench exit $?"; grep -v amdgpu.ids gpurun_out/bench_bits.err | tail -8; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_bits.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
for v in d.get("variants",[]): print(v["name"], "|", v["kernel"], v["kernel_avg_us"], v.get("gteps_local"), v.get("identical_to_csr_kernel"))
PY
