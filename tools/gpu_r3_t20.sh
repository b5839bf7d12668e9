#!/bin/bash
# round 3: hub ranges finished by finisher workgroups at the end of the main launch (SPLAHIP_COB_HUB_FINISH=2) against the second launch (1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity, finishers"; SPLAHIP_COB_HUB_FINISH=2 timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
  for sc in 22 23; do
    for hf in 2 1 2 1; do echo "== scale $sc hub_finish $hf"; SPLAHIP_COB_HUB_FINISH=$hf EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|min-plus" | cut -c1-60; done
  done
  for fs in 1280 5120; do echo "== scale 22 hub_finish 2 fin_slots $fs"; SPLAHIP_COB_FIN_SLOTS=$fs SPLAHIP_COB_HUB_FINISH=2 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py 22 0:0 2>&1 | grep "^cob\|min-plus" | cut -c1-60; done
  echo "== configs, finishers"; SPLAHIP_COB_HUB_FINISH=2 timeout 2400 python -m pytest tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
} 2>&1 | tee gpurun_out/r3_t20.txt
