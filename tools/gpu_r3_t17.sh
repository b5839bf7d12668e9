#!/bin/bash
# round 3: natural ranges of 32704 slots (fp32 sums of natural rows in float words, compare-and-swap adds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 2400 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8
  for sc in 22 23; do
    for ns in 32704 20224; do echo "== scale $sc natural slots $ns"; SPLAHIP_COB_NAT_SLOTS=$ns SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|schedule\|min-plus"; done
  done
  echo "== configs"; timeout 2400 python -m pytest tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8
} 2>&1 | tee gpurun_out/r3_t17.txt
