#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
timeout 300 python -m pytest tests/test_mxv_gpu.py -x -q --timeout 120 2>&1 | grep -E "passed|failed|Error|error|core" | tail -5 > gpurun_out/r04_t14.txt
timeout 200 python tools/exp/exp_r4_mask.py 23 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_t14.txt
cd /tmp; export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_mask"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_mask" -- python3 "$R/tools/exp/exp_r4_mask.py" 23 > /dev/null 2>&1
cd "$R"; cat gpurun_out/r04_t14.txt
