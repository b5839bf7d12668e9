#!/bin/bash
# round 4: the persistent form with the pipeline kept running across pieces: parity, schedule grid, timeline
out=gpurun_out/r04_t6.txt
: > $out
timeout 900 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $out
for cfg in "90 3 0" "70 4 0" "60 4 0" "70 5 0" "50 5 0" "90 4 1" "70 4 1"; do
  set -- $cfg
  timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_SHARE_PCT=$1 SPLAHIP_COB_LEVELS=$2 SPLAHIP_COB_TAPER=$3 EXP_CHECK=$([ "$1" = "90" ] && echo 1 || echo 0) 2>&1 | grep -v amdgpu.ids | cut -c1-190 >> $out
done
for cfg in "110 3 0" "70 4 0" "70 5 0"; do
  set -- $cfg
  timeout 300 python tools/exp/exp_r4_cob.py 22 0 SPLAHIP_COB_SHARE_PCT=$1 SPLAHIP_COB_LEVELS=$2 SPLAHIP_COB_TAPER=$3 EXP_CHECK=0 2>&1 | grep -v amdgpu.ids | cut -c1-190 >> $out
done
run() { echo "== $*" >> $out; env "$@" SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 300 python tools/exp/exp_cob_timing.py 23 2>&1 | grep -v "amdgpu.ids\|schedule 0" >> $out; }
run SPLAHIP_COB_SHARE_PCT=70 SPLAHIP_COB_LEVELS=5
cat $out
