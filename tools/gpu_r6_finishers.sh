#!/bin/bash
# Round 6: the hub ranges' row pass by finisher workgroups inside the main launch (SPLAHIP_COB_HUB_FINISH=2) — first a short guarded run, then the tests, then A/B timings
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 120 python3 -m pytest tests/test_mxv_gpu.py -x -q -m gpu --timeout 100 -k "test_all_semirings or naive or structural" 2>&1 | tail -3 || exit 1
timeout 1200 python3 -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py tests/test_golden.py -x -q -m gpu --timeout 600 2>&1 | tail -4
for m in 1 2; do
  echo "== SPLAHIP_COB_HUB_FINISH=$m"
  for sc in 23 22; do SPLAHIP_COB_HUB_FINISH=$m timeout 300 python3 bench.py --scale $sc --steps 50 --warmup 5 --no-cpu-baseline --no-verify --no-variants 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  scale', $sc, 'ms_per_step', d['ms_per_step'], 'kernel_avg_us', d['roofline']['kernel_avg_us'], 'frac', d['roofline']['frac'])"; done
done
