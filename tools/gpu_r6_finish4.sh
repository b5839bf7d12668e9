#!/bin/bash
# Round 6: the hub ranges' finishing kernel with four slots per thread (SPLAHIP_COB_FINISH4=1) against one (0): tests, the headline, the PageRank and SSSP steps
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_finish4.txt
: > $out
timeout 1200 python3 -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -x -q -m gpu --timeout 600 2>&1 | tail -2 | tee -a $out
for f in 1 0 1 0; do
  for sc in 23 22; do
    echo -n "SPLAHIP_COB_FINISH4=$f scale $sc: " | tee -a $out
    SPLAHIP_COB_FINISH4=$f timeout 300 python3 bench.py --scale $sc --steps 50 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel_avg_us', d['roofline']['kernel_avg_us'], 'frac', d['roofline']['frac'], 'verified', d.get('verified'))" | tee -a $out
  done
done
