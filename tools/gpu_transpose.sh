#!/bin/bash
# transpose / map parity + the transpose workload line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --workload transpose --steps 10 --warmup 2 > gpurun_out/ops_transpose.json 2> gpurun_out/ops_transpose.err; echo "exit $?"; grep -v amdgpu.ids gpurun_out/ops_transpose.err | tail -5; cat gpurun_out/ops_transpose.json
timeout 600 python bench.py --workload pagerank-vxm --steps 10 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -3
