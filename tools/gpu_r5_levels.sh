#!/bin/bash
# Round 5: four size levels of the pieces (SPLAHIP_COB_LEVELS=4) against the shipped three, shares 70 - 100 % and the model's own pick, R-MAT 23 / 22 / 24, two rounds.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_levels.txt
: > $out
for s in 23 22 24; do
  for rep in 1 2; do
    for kv in "X=1" "SPLAHIP_COB_LEVELS=4" "SPLAHIP_COB_LEVELS=4 SPLAHIP_COB_SHARE_PCT=70" "SPLAHIP_COB_LEVELS=4 SPLAHIP_COB_SHARE_PCT=80" "SPLAHIP_COB_LEVELS=4 SPLAHIP_COB_SHARE_PCT=90" "SPLAHIP_COB_LEVELS=4 SPLAHIP_COB_SHARE_PCT=100"; do
      echo -n "scale $s [$kv]: " >> $out
      env $kv timeout 300 python3 tools/exp/exp_r4_ab.py $s 2>&1 | grep "tag=" | cut -c14-120 >> $out
    done
  done
done
