#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py --workload algorithms --steps 5 --warmup 1 > gpurun_out/ops_algorithms.json 2> gpurun_out/ops_algorithms.err; echo "exit $?"; grep -v amdgpu.ids gpurun_out/ops_algorithms.err | tail -12
