#!/bin/bash
# quick GPU session: parity tests + experiment + bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 600 python tools/exp/exp_gather.py 22 --sell 2>&1 | grep -v amdgpu.ids
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; grep -v amdgpu.ids gpurun_out/bench.err | tail; cat gpurun_out/bench.json
