#!/bin/bash
# round 3: the LDS slice set to the identity under the prologue's loads (new build) against the build before (SPLA_LIB_TAG=prev)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
  for sc in 22 23; do
    for tag in new prev new prev new prev; do echo "== scale $sc build $tag"; if [ $tag = prev ]; then export SPLA_LIB_TAG=prev; else unset SPLA_LIB_TAG; fi; EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|min-plus" | cut -c1-70; done
  done
  unset SPLA_LIB_TAG
} 2>&1 | tee gpurun_out/r3_t32.txt
