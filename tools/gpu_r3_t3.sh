#!/bin/bash
# round 3: cob v2 with batched row passes: parity (mxv, vxm, shard, full-size cob tests), sweep of the schedule knobs, timeline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | tail -6
  timeout 1500 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "column_ordered" 2>&1 | tail -6
  for sc in 22 23; do
    echo "== scale $sc"; SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0:50:25 0:0:35:25 0:0:70:25 0:0:100:25 0:0:50:15 0:0:50:40 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
  for sc in 22 23; do echo "== timeline $sc"; SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v amdgpu.ids; done
} 2>&1 | tee gpurun_out/r3_t3.txt
