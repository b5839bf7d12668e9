#!/bin/bash
# round 3: cob v2 (hub + natural ranges, in-kernel merge): parity suites that touch it, then timings
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | tail -15
  for sc in 22 23; do
    echo "== scale $sc"; SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 600 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep -v amdgpu.ids
  done
} 2>&1 | tee gpurun_out/r3_t1.txt
