#!/bin/bash
# Round 5: whole natural ranges dealt to XCDs in groups of similar length (SPLAHIP_COB_NAT_XCD=1) against round-robin, R-MAT 23 / 22 / 24.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_natxcd.txt
: > $out
for s in 23 22 24; do
  for c in 1 0 1 0; do
    echo -n "scale $s nat_xcd=$c: " >> $out
    SPLAHIP_COB_NAT_XCD=$c timeout 300 python3 tools/exp/exp_r4_ab.py $s 2>&1 | grep "tag=" >> $out
  done
done
