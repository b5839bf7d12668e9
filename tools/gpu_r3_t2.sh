#!/bin/bash
# round 3: cob v2 with cost-cut pieces: parity, then a sweep of the schedule knobs, then the timeline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 1500 python -m pytest tests/test_mxv_gpu.py -m gpu -x -q 2>&1 | tail -15
  for sc in 22 23; do
    echo "== scale $sc"; SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0:50:60 0:0:50:40 0:0:50:100 0:0:30:60 0:0:80:60 0:0:100:60 0:0:50:60:8 0:0:50:60:32 0:0:50:60:0 2>&1 | grep -v amdgpu.ids
  done
  for sc in 22 23; do echo "== timeline $sc"; SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v amdgpu.ids; done
} 2>&1 | tee gpurun_out/r3_t2.txt
