#!/bin/bash
# round 4: schedule grid of the persistent form (share %, levels, taper), R-MAT 23 and 22
out=gpurun_out/r04_t5.txt
: > $out
for cfg in "90 3 0" "50 5 0" "60 5 0" "70 5 0" "50 4 0" "60 4 0" "40 5 0" "50 6 0" "50 4 1" "70 4 1" "90 4 1" "110 5 1"; do
  set -- $cfg
  timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_SHARE_PCT=$1 SPLAHIP_COB_LEVELS=$2 SPLAHIP_COB_TAPER=$3 EXP_CHECK=0 2>&1 | grep -v amdgpu.ids >> $out
done
for cfg in "110 3 0" "50 5 0" "70 5 0" "50 4 0" "70 4 1"; do
  set -- $cfg
  timeout 300 python tools/exp/exp_r4_cob.py 22 0 SPLAHIP_COB_SHARE_PCT=$1 SPLAHIP_COB_LEVELS=$2 SPLAHIP_COB_TAPER=$3 EXP_CHECK=0 2>&1 | grep -v amdgpu.ids >> $out
done
cut -c1-200 $out
