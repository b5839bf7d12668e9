#!/bin/bash
# round 3: rows a natural range may span (x its slots): 8 (default) / 4 / 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for sc in 22 23 24; do
    for cap in 8 4 2 8 4 2; do echo "== scale $sc span cap $cap"; SPLAHIP_COB_SPAN_CAP=$cap EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|min-plus" | cut -c1-70; done
  done
} 2>&1 | tee gpurun_out/r3_t30.txt
