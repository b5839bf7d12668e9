#!/bin/bash
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_mxv_gpu.py -x -q -m gpu --timeout 600 -k "selected_rows or masked_call" 2>&1 | tail -12
cd $GRAFT_REPO_ROOT
echo "== default weights"; timeout 300 python3 tools/exp/exp_r4_mask.py 23 2>&1 | grep -v "^\[cob\]\|amdgpu.ids"
echo "== rows forced"; timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_MASK_ROWS_X10=1 SPLAHIP_MASK_LONG_X10=1 2>&1 | grep -v "^\[cob\]\|amdgpu.ids"
MASKS="random10 random50" X10=1 bash tools/gpu_r6_mask2.sh
