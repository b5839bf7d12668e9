#!/bin/bash
# round 4: guided self-scheduling grid (k x 10, minimum chunks, order), R-MAT 23 / 22
out=gpurun_out/r04_t7.txt
: > $out
for cfg in "15 128 0" "10 128 0" "20 128 0" "30 128 0" "15 64 0" "15 256 0" "15 128 1" "10 128 1" "20 256 1" "20 64 0"; do
  set -- $cfg
  timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_GSS_K10=$1 SPLAHIP_COB_GSS_MIN=$2 SPLAHIP_COB_GSS_ORDER=$3 EXP_CHECK=$([ "$cfg" = "15 128 0" ] && echo 1 || echo 0) 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $out
done
for cfg in "15 128 0" "10 128 0" "20 64 0" "15 128 1"; do
  set -- $cfg
  timeout 300 python tools/exp/exp_r4_cob.py 22 0 SPLAHIP_COB_GSS_K10=$1 SPLAHIP_COB_GSS_MIN=$2 SPLAHIP_COB_GSS_ORDER=$3 EXP_CHECK=0 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $out
done
cat $out
