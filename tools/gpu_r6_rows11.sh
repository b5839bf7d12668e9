#!/bin/bash
# Round 6: the segment-item tasks beside the compaction pass (SPLAHIP_ROWS_ITEMS_EARLY=1) against behind it (0)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows11.txt
: > $out
SPLAHIP_ROWS_ITEMS_EARLY=1 timeout 600 python3 -m pytest tests/test_mxv_gpu.py -x -q -m gpu --timeout 500 -k "selected_rows or masked" 2>&1 | grep -a "passed\|failed" | tail -1 | tee -a $out
for f in 1 0 1 0; do
  echo "== SPLAHIP_ROWS_ITEMS_EARLY=$f" | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_ITEMS_EARLY=$f 2>&1 | grep "random 10\|no hub" | sed 's/selected nnz *[0-9.]* M ([0-9.]*) *//' | cut -c1-70 | tee -a $out
done
