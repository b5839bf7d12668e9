#!/bin/bash
# column-ordered-blocks kernel: where the time goes (diagnostic builds -DSPLAHIP_COB_EXP=n: 1 no LDS accumulation, 2 no gathers, 3 plain LDS stores,
# 4 = 2 + 3, 5 ds_add_f32 instead of the compare-and-swap sum)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 600 python -m pytest tests/test_mxv_gpu.py -m gpu -x -q 2>&1 | tail -3
  echo "== product build"; EXP_INT=1 EXP_SELL=0 python tools/exp/exp_cob.py 22 ${COB_CFGS:-14:256 15:256 13:128} 2>&1 | grep -v amdgpu.ids
  for v in ${COB_EXPS:-1 3 4}; do echo "== SPLAHIP_COB_EXP=$v"; EXP_SELL=0 SPLA_LIB_TAG=cobexp$v python tools/exp/exp_cob.py 22 14:256 2>&1 | grep "^cob"; done
  rm -rf gpurun_out/prof_cob; EXP_INT=1 EXP_SELL=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cob --output-format csv -- python tools/exp/exp_cob.py 22 14:256 > /dev/null 2>&1
  f=$(ls gpurun_out/prof_cob/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== rocprof kernel stats ($f)"; head -12 "$f" | cut -c1-200
} 2>&1 | tee gpurun_out/cob_exp.txt
