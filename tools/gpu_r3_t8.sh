#!/bin/bash
# round 3: per-lane spare words for padding slots; lazy sliced-ELL copy; sparse exchange
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py tests/test_host_api.py -m gpu -x -q 2>&1 | tail -4
  for sc in 22 23; do
    for lv in 0 1 3; do echo "== scale $sc levels $lv"; SPLAHIP_COB_LEVELS=$lv SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1\|min-plus" | sort -u; done
    echo "== scale $sc slots 10112"; SPLAHIP_COB_SLOTS=10112 SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1"
    echo "== timeline $sc"; SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
} 2>&1 | tee gpurun_out/r3_t8.txt
