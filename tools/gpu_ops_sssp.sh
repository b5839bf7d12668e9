#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
timeout 900 python bench.py --workload sssp-step --steps 10 --warmup 2 > gpurun_out/ops_sssp-step.json 2> gpurun_out/ops_sssp-step.err; echo "exit $?"; grep -v amdgpu.ids gpurun_out/ops_sssp-step.err | tail -4; cut -c1-600 gpurun_out/ops_sssp-step.json
SPLA_BENCH_PLAN=csr timeout 900 python bench.py --workload sssp-step --steps 10 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | cut -c1-200
