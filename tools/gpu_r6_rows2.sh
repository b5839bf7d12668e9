#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_mxv_gpu.py -x -q -m gpu --timeout 600 -k "selected_rows or masked_call" 2>&1 | tail -3
for lb in 2048 4096 8192; do
echo "== rows forced, task blocks $lb"; timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_MASK_ROWS_X10=1 SPLAHIP_MASK_LONG_X10=1 SPLAHIP_ROWS_LONG_BLOCKS=$lb 2>&1 | grep -v "^\[cob\]\|amdgpu.ids"
done
MASKS="random10" X10=1 bash tools/gpu_r6_mask2.sh
