#!/bin/bash
# Round 6: column slices with XCD affinity (SPLAHIP_COB_XCD_SLICES = 8 / 16 / 32 against 0) — a quick parity run with the slices on, then A/B timings at R-MAT 23 / 22 / 24
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_slices.txt
: > $out
SPLAHIP_COB_XCD_SLICES=8 timeout 900 python3 -m pytest tests/test_mxv_gpu.py tests/test_configs_gpu.py -x -q -m gpu --timeout 600 -k "not soak" 2>&1 | tail -4 | tee -a $out
for sc in 23 22 24; do
  for m in 0 8 16 0 8 32; do
    echo -n "scale $sc slices=$m: " | tee -a $out
    SPLAHIP_COB_TRACE=1 SPLAHIP_COB_XCD_SLICES=$m timeout 300 python3 bench.py --scale $sc --steps 50 --warmup 5 --no-cpu-baseline --no-variants 2> gpurun_out/r6_slices_err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel_avg_us', d['roofline']['kernel_avg_us'], 'frac', d['roofline']['frac'], 'verified', d.get('verified'))" | tee -a $out
    grep "schedule 1" gpurun_out/r6_slices_err.txt | tail -1 | cut -c1-330 | tee -a $out
  done
done
