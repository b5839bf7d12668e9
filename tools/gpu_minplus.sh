#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mxv_gpu.py tests/test_api_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -6
timeout 300 python tools/exp/fuzz_mxv.py 60 31 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py --workload sssp-step --steps 10 --warmup 2 > gpurun_out/ops_sssp-step.json 2> gpurun_out/ops_sssp-step.err; echo "exit $?"; grep -v amdgpu.ids gpurun_out/ops_sssp-step.err | tail -3; cut -c1-240 gpurun_out/ops_sssp-step.json
SPLAHIP_SELL_BITS=0 timeout 900 python bench.py --workload sssp-step --steps 10 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | cut -c1-200
