#!/bin/bash
# Round 6: workgroup counts of the selected-rows launches after the fusions (random 10 % mask, R-MAT 23)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows7.txt
: > $out
for cfg in "1024 4096 4096" "768 4096 4096" "896 4096 4096" "640 4096 4096" "1024 2048 4096" "1024 8192 4096" "1024 4096 8192" "1024 4096 2048" "1024 3072 6144" "1024 4096 4096"; do
  set -- $cfg
  echo -n "compact $1 short $2 task $3: " | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_COMPACT_BLOCKS=$1 SPLAHIP_ROWS_SHORT_BLOCKS=$2 SPLAHIP_ROWS_LONG_BLOCKS=$3 2>&1 | grep "random 10\|50 %, no hub" | cut -c1-90 | tr '\n' ' ' | tee -a $out; echo | tee -a $out
done
