#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_user_ops_gpu.py tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_golden.py tests/test_api_gpu.py -x -q -m gpu --timeout 900 2>&1 | tail -8
timeout 600 python3 tools/exp/exp_r6_ordered_ops.py 23 2>&1 | grep -v amdgpu.ids
SPLAHIP_FORCE_ROWSEQ=1 timeout 900 python3 tools/exp/exp_r6_ordered_ops.py 22 2>&1 | grep -v amdgpu.ids
timeout 900 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu --timeout 900 -k "reference_order" -s 2>&1 | grep -a "reference order\|passed\|failed"
