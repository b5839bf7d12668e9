#!/bin/bash
# round 3: equal cost per size level (SPLAHIP_COB_TAPER=1) against equal counts (0), with three and four levels; natural pieces first
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for sc in 22 23; do
    for cfg in "0 3" "1 3" "1 4" "1 2" "0 3" "1 3" "1 4" "1 2"; do set -- $cfg; echo "== scale $sc taper $1 levels $2"; SPLAHIP_COB_TAPER=$1 SPLAHIP_COB_LEVELS=$2 SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1" | cut -c1-330 | sed 's/^\[cob\] schedule 1.*piece <=/  model: piece <=/' | cut -c1-150; done
  done
} 2>&1 | tee gpurun_out/r3_t26.txt
