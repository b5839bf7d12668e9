#!/bin/bash
# round 3: int32 on ranges of 32704 slots (one workgroup per CU) against 20224 (two): what do larger ranges give?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for sc in 22 23; do
    for R in 20224 32704; do echo "== scale $sc slots $R"; SPLAHIP_COB_SLOTS=$R SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 0\|min-plus"; done
  done
} 2>&1 | tee gpurun_out/r3_t16.txt
