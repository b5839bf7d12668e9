#!/bin/bash
# round 3: static op_add in the row pass + all-empty fast path; hub finish on by default; then the bench line and the two-rank control-flow runs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== parity"; timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py tests/test_mxm_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
  for sc in 22 23; do
    for hf in 1 0; do echo "== scale $sc hub_finish $hf"; SPLAHIP_COB_HUB_FINISH=$hf SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 0:0:80:25 2>&1 | grep "^cob\|schedule 1\|min-plus" | sort -u; done
    echo "== timeline $sc"; SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
  echo "== bench.py"; timeout 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.log; tail -12 gpurun_out/r03_bench.log; head -c 2500 gpurun_out/r03_bench.json; echo
  echo "== n2 debug"; timeout 1200 bash tools/gpu_n2_debug_r03.sh 2>&1 | tail -30
} 2>&1 | tee gpurun_out/r3_t12.txt
