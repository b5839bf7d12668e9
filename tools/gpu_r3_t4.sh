#!/bin/bash
# round 3: cob v2 with model-chosen piece size, 16-byte accumulator read-back: full-size parity, sweep, timeline, the new default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 1500 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py -m gpu -x -q 2>&1 | tail -4
  timeout 1500 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "column_ordered" 2>&1 | tail -6
  for sc in 22 23; do
    echo "== scale $sc"; SPLAHIP_COB_TRACE=1 EXP_INT=1 EXP_SELL=0 timeout 900 python tools/exp/exp_cob.py $sc 0:0 0:0:60:25 0:0:45:25 0:0:80:25 2>&1 | grep -v "amdgpu.ids\|schedule 0"
  done
  for sc in 22 23; do echo "== timeline $sc"; SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"; done
  echo "== bench.py"; timeout 900 python bench.py > gpurun_out/r3_bench_first.json 2> gpurun_out/r3_bench_first.log; tail -25 gpurun_out/r3_bench_first.log; head -c 3000 gpurun_out/r3_bench_first.json
} 2>&1 | tee gpurun_out/r3_t4.txt
