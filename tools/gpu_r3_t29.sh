#!/bin/bash
# round 3: timelines of the shipped schedule (timing build of the final sources); the bench line at other sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for sc in 22 23; do echo "== timeline $sc"; SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 600 python tools/exp/exp_cob_timing.py $sc 2>&1 | grep -v "amdgpu.ids\|schedule 0"; done
  for sc in 20 21 24; do echo "== bench --scale $sc"; timeout 900 python bench.py --scale $sc --no-cpu-baseline --no-variants 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d[k] for k in ('value','ms_per_step','verified')}), d['roofline']['frac'], d['roofline']['kernel_avg_us'], d['config']['n'], d['config']['nnz'], d['config']['layout'])"; done
} 2>&1 | tee gpurun_out/r3_t29.txt
