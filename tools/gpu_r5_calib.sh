#!/bin/bash
# Round 5: the calibrated launch order (SPLAHIP_COB_CALIBRATE) against the order as cut, R-MAT 22/23/24, and the blocks' parity tests.
export TMPDIR=/tmp
out=gpurun_out/r5_calib.txt
: > $out
for s in 23 22 24; do
  for c in 1 0 1 0; do
    echo "== scale $s calibrate=$c" >> $out
    SPLAHIP_COB_TRACE=$([ $c = 1 ] && echo 1 || echo "") SPLAHIP_COB_CALIBRATE=$c timeout 300 python3 tools/exp/exp_r4_ab.py $s 2>&1 | grep -v "^\[cob\] \(attach\|sched\|build\)" | tail -12 >> $out
  done
done
timeout 900 python3 -m pytest tests -x -q -m gpu -k "cob or blocks" --timeout 600 2>&1 | tail -5 >> $out
