#!/bin/bash
# Round 6: the probe inside the row kernels' first launch: LDS per workgroup decides whether every workgroup of the launch is resident at once
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows6.txt
: > $out
for cfg in "1 1024" "0 1024" "1 1024" "0 896" "1 1280"; do
  set -- $cfg
  echo "== SPLAHIP_ROWS_SMALL_LDS=$1 SPLAHIP_ROWS_COMPACT_BLOCKS=$2" | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_SMALL_LDS=$1 SPLAHIP_ROWS_COMPACT_BLOCKS=$2 2>&1 | grep "all rows (\|random 10\|no hub" | cut -c1-100 | tee -a $out
done
