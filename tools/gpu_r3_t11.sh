#!/bin/bash
# round 3: the hub ranges' row pass as a kernel of its own (SPLAHIP_COB_HUB_FINISH=1) against the last-arriver form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity with the finishing kernel"; SPLAHIP_COB_HUB_FINISH=1 timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
  for sc in 22 23; do
    for hf in 0 1; do echo "== scale $sc hub_finish $hf"; SPLAHIP_COB_HUB_FINISH=$hf SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 0:0:100:25 0:0:80:25 2>&1 | grep "^cob\|schedule 1\|min-plus" | sort -u; done
    echo "== timeline $sc hub_finish 1"; SPLAHIP_COB_HUB_FINISH=1 SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
} 2>&1 | tee gpurun_out/r3_t11.txt
