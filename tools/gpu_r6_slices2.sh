#!/bin/bash
# Round 6: column slices with XCD affinity, the hub ranges only / the natural ranges only (SPLAHIP_COB_XCD_WHAT = 1 / 2), R-MAT 23
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_slices2.txt
: > $out
for cfg in "0 3" "8 1" "8 2" "0 3" "8 1" "16 1" "4 1"; do
  set -- $cfg
  echo -n "scale 23 slices=$1 what=$2: " | tee -a $out
  SPLAHIP_COB_TRACE=1 SPLAHIP_COB_XCD_SLICES=$1 SPLAHIP_COB_XCD_WHAT=$2 timeout 300 python3 bench.py --scale 23 --steps 50 --warmup 5 --no-cpu-baseline --no-variants 2> gpurun_out/r6_slices_err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel_avg_us', d['roofline']['kernel_avg_us'], 'frac', d['roofline']['frac'], 'verified', d.get('verified'))" | tee -a $out
  grep "\] schedule 1" gpurun_out/r6_slices_err.txt | tail -1 | cut -c1-330 | tee -a $out
done
