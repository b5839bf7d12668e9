#!/bin/bash
# round 3, first session: machine model for the cob kernel (stream ceiling, gather cost by residency), diagnostic builds, per-workgroup timeline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== stream_bench"; timeout 300 tools/exp/bin/stream_bench
  for sc in 22 23; do
    echo "== product build, scale $sc"; EXP_SELL=0 timeout 600 python tools/exp/exp_cob.py $sc 14:0 2>&1 | grep -v amdgpu.ids
    for v in 2 16 20; do echo "== SPLAHIP_COB_EXP=$v scale $sc"; EXP_SELL=0 SPLA_LIB_TAG=cobexp$v timeout 600 python tools/exp/exp_cob.py $sc 14:0 2>&1 | grep "^cob"; done
    echo "== timeline, scale $sc"; SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v amdgpu.ids
  done
} 2>&1 | tee gpurun_out/r3_exp1.txt
