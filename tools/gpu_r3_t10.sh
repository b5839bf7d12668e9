#!/bin/bash
# round 3: launch order experiment (range-major), hub rows requested by the last arriver only
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for sc in 22 23; do
    for ord in 0 1; do for lv in 1 3; do echo "== scale $sc order $ord levels $lv"; SPLAHIP_COB_ORDER=$ord SPLAHIP_COB_LEVELS=$lv SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 0:0:50:25 0:0:100:25 2>&1 | grep "^cob\|schedule 1" | sort -u; done; done
    echo "== timeline $sc order 1"; SPLAHIP_COB_ORDER=1 SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
} 2>&1 | tee gpurun_out/r3_t10.txt
