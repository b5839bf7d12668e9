#!/bin/bash
# round 3: pieces cut at whole rounds of the loop (64 chunks) against pieces as the cost cut them
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
  for sc in 22 23 24; do
    for rc in 64 0 64 0 64 0; do echo "== scale $sc round chunks $rc"; SPLAHIP_COB_ROUND_CHUNKS=$rc EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 2>&1 | grep "^cob\|min-plus" | cut -c1-70; done
  done
} 2>&1 | tee gpurun_out/r3_t31.txt
