#!/bin/bash
# round 3: piece-end function with early hub-row loads; full GPU suite; bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for sc in 22 23; do
    for lv in 0 3; do echo "== scale $sc levels $lv"; SPLAHIP_COB_LEVELS=$lv SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1\|min-plus" | sort -u; done
    echo "== timeline $sc"; SPLAHIP_COB_LEVELS=3 SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
  echo "== full gpu suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15
} 2>&1 | tee gpurun_out/r3_t9.txt
