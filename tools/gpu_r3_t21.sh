#!/bin/bash
# round 3: more hub ranges at R-MAT 23 (16 ranges do not hold every row of degree >= 2 x average there), gather-line weight of the cost model
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== scale 23 hub ranges 16 / 22 / 28"; SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 1200 python tools/exp/exp_cob.py 23 0:0:0:25:16 0:0:0:25:22 0:0:0:25:28 2>&1 | grep "^cob\|schedule 1\|min-plus" | cut -c1-330
  echo "== scale 23 hub ranges 22, natural slots 32704"; SPLAHIP_COB_NAT_SLOTS=32704 SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py 23 0:0:0:25:22 2>&1 | grep "^cob\|schedule 1\|min-plus" | cut -c1-330
  for sc in 22 23; do
    echo "== scale $sc line weight 15 / 25 / 40"; EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0:0:15 0:0:0:25 0:0:0:40 2>&1 | grep "^cob" | cut -c1-100
  done
  echo "== scale 24 hub ranges 16 / 32 (int32)"; SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 1200 python tools/exp/exp_cob.py 24 0:0:0:25:16 0:0:0:25:32 2>&1 | grep "^cob\|schedule 1\|min-plus" | cut -c1-330
} 2>&1 | tee gpurun_out/r3_t21.txt
