#!/bin/bash
# round 3: noinline row pass, tapered pieces, JIT for vxm / mxm / mxmT: parity suites, sweep, timeline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py tests/test_user_ops_gpu.py -m gpu -x -q 2>&1 | tail -8
  timeout 1500 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "column_ordered" 2>&1 | tail -6
  for sc in 22 23; do
    echo "== scale $sc"; SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep -v "amdgpu.ids\|schedule 0"
    for lv in 1 2 3; do SPLAHIP_COB_LEVELS=$lv SPLAHIP_COB_TRACE=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule 1" | sed "s/^/levels $lv: /"; done
  done
  for sc in 22 23; do echo "== timeline $sc"; SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"; done
} 2>&1 | tee gpurun_out/r3_t5.txt
