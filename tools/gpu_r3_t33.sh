#!/bin/bash
# round 3: the bit forms' kernels without the 64-register cap (SPLAHIP_COB_MIN_WAVES=4 build, SPLA_LIB_TAG=mw4) against the shipped build; parity of the rebuilt default
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity (default build)"; timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
  for tag in default mw4 default mw4; do
    if [ $tag = mw4 ]; then export SPLA_LIB_TAG=mw4; else unset SPLA_LIB_TAG; fi
    echo "== sssp-step R-MAT 24, build $tag"; timeout 900 python bench.py --workload sssp-step --steps 20 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_avg_us'), d['verified'])"
    echo "== scale 23 int32 (bit form), build $tag"; EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py 23 0:0 2>&1 | grep "^cob\|min-plus" | cut -c1-70
  done
  unset SPLA_LIB_TAG
  echo "== parity (mw4 build)"; SPLA_LIB_TAG=mw4 timeout 1500 python -m pytest tests/test_mxv_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3
} 2>&1 | tee gpurun_out/r3_t33.txt
