#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_vxm_gpu.py tests/test_algorithms.py tests/test_vector_ops.py tests/test_golden_next.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/exp/exp_bfs_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -7
timeout 900 python bench.py --workload algorithms --steps 5 --warmup 1 > gpurun_out/ops_algorithms.json 2> gpurun_out/ops_algorithms.err; echo "exit $?"; grep -v amdgpu.ids gpurun_out/ops_algorithms.err | tail -7
