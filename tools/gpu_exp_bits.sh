#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
timeout 900 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/exp/exp_bits.py base 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/exp/exp_sell_min.py fp32-all-rows 2>&1 | grep -v amdgpu.ids | tail -1
