#!/bin/bash
# Round 6: workgroups of the wave-task and short-row kernels on the final build (random 10 % mask, R-MAT 23), three repeats
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_rows8.txt
: > $out
for rep in 1 2 3; do
for cfg in "4096 4096" "4096 8192" "8192 8192" "4096 12288" "2048 8192"; do
  set -- $cfg
  echo -n "rep $rep short $1 task $2: " | tee -a $out
  timeout 300 python3 tools/exp/exp_r4_mask.py 23 SPLAHIP_ROWS_SHORT_BLOCKS=$1 SPLAHIP_ROWS_LONG_BLOCKS=$2 2>&1 | grep "random 10\|50 %, no hub" | sed 's/selected nnz *[0-9.]* M ([0-9.]*) *//' | cut -c1-60 | tr '\n' ' ' | tee -a $out; echo | tee -a $out
done
done
