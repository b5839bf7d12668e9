#!/bin/bash
# Round 6: the selected-rows path with independent passes sharing a launch (SPLAHIP_ROWS_FUSED=1: select | compact, short rows | tasks) against four launches (0)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows3.txt
: > $out
timeout 900 python3 -m pytest tests/test_mxv_gpu.py -x -q -m gpu --timeout 600 -k "selected_rows or masked" 2>&1 | tail -3 | tee -a $out
for f in 1 0 1 0; do
  echo "== SPLAHIP_ROWS_FUSED=$f" | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_FUSED=$f 2>&1 | grep -v "^\[cob\]\|amdgpu.ids" | tee -a $out
done
timeout 1200 python3 -m pytest tests/test_mxv_gpu.py tests/test_configs_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -2 | tee -a $out
