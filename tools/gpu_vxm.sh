#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vxm_gpu.py tests/test_mxv_gpu.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py --workload pagerank-vxm --steps 10 --warmup 2 > gpurun_out/ops_pagerank-vxm.json 2> gpurun_out/ops_pagerank-vxm.err; echo "exit $?"; grep -v amdgpu.ids gpurun_out/ops_pagerank-vxm.err | tail -3; cut -c1-260 gpurun_out/ops_pagerank-vxm.json
SPLA_BENCH_PLAN=csr timeout 900 python bench.py --workload pagerank-vxm --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-200
