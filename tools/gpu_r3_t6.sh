#!/bin/bash
# round 3: two 8-wave workgroups per CU (R = 10112) against one 16-wave workgroup (R = 20224) for the fp32 PLUS form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py -m gpu -x -q 2>&1 | tail -4
  timeout 1500 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "cfg3_full" 2>&1 | tail -4
  for sc in 22 23; do
    for slots in 20224 10112; do
      for lv in 0 1 3; do echo "== scale $sc slots $slots levels $lv"; SPLAHIP_COB_SLOTS=$slots SPLAHIP_COB_LEVELS=$lv SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1"; done
    done
    echo "== timeline $sc slots 10112"; SPLAHIP_COB_SLOTS=10112 SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
} 2>&1 | tee gpurun_out/r3_t6.txt
