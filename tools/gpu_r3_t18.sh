#!/bin/bash
# round 3: timelines of the fp32 call with natural ranges of 32704 / 20224 slots
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for ns in 32704 20224; do
    echo "== timeline 22 natural slots $ns"; SPLAHIP_COB_NAT_SLOTS=$ns SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py 22 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
  for sc in 22 23 24; do
    for pc in 1 2; do echo "== scale $sc int32, natural slots 20224, workgroups per CU $pc"; SPLAHIP_COB_PER_CU=$pc SPLAHIP_COB_NAT_SLOTS=20224 SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 EXP_FP32=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "schedule 0\|min-plus"; done
  done
} 2>&1 | tee gpurun_out/r3_t18.txt
