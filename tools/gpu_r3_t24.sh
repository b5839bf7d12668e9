#!/bin/bash
# round 3: timelines of the shipped defaults; mxm at R-MAT 19 on this round's code
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
  for sc in 22 23; do echo "== timeline $sc"; SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"; done
  echo "== mxm scale 19"; timeout 1500 python bench.py --workload mxm --scale 19 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/r03/ops_mxm_scale19.json 2> gpurun_out/r03/ops_mxm_scale19.err; echo "exit $?"; cut -c1-600 gpurun_out/r03/ops_mxm_scale19.json; tail -3 gpurun_out/r03/ops_mxm_scale19.err
} 2>&1 | tee gpurun_out/r3_t24.txt
