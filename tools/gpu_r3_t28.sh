#!/bin/bash
# round 3: timelines with natural pieces first (the shipped order)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{ for sc in 22 23; do echo "== timeline $sc"; SPLAHIP_COB_ORDER=2 SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"; done; } 2>&1 | tee gpurun_out/r3_t28.txt
