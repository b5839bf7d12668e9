#!/bin/bash
# Round 6: the compaction pass of the selected-rows path with the next step's words requested ahead; LDS buffers of 26 KB instead of 37 (SPLAHIP_ROWS_SMALL_LDS=1) x workgroups
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows4.txt
: > $out
timeout 900 python3 -m pytest tests/test_mxv_gpu.py -x -q -m gpu --timeout 600 -k "selected_rows or masked" 2>&1 | tail -3 | tee -a $out
for cfg in "0 1024" "1 1024" "1 1536" "1 2048" "0 2048" "0 1024"; do
  set -- $cfg
  echo "== SPLAHIP_ROWS_SMALL_LDS=$1 SPLAHIP_ROWS_COMPACT_BLOCKS=$2" | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_SMALL_LDS=$1 SPLAHIP_ROWS_COMPACT_BLOCKS=$2 2>&1 | grep "random 10\|no hub" | cut -c1-100 | tee -a $out
done
