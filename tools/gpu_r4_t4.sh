#!/bin/bash
# round 4: persistent form with deferred tickets: parity, timing, timeline
out=gpurun_out/r04_t4.txt
: > $out
timeout 900 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -x -q 2>&1 | tail -5 >> $out
for lv in 3 4 5; do
  timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_SHARE_PCT=90 SPLAHIP_COB_LEVELS=$lv >> $out 2>&1
done
timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_SHARE_PCT=110 SPLAHIP_COB_LEVELS=4 >> $out 2>&1
timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_SHARE_PCT=70 SPLAHIP_COB_LEVELS=3 >> $out 2>&1
timeout 300 python tools/exp/exp_r4_cob.py 22 0 SPLAHIP_COB_SHARE_PCT=110 SPLAHIP_COB_LEVELS=3 >> $out 2>&1
timeout 300 python tools/exp/exp_r4_cob.py 22 0 SPLAHIP_COB_SHARE_PCT=110 SPLAHIP_COB_LEVELS=4 >> $out 2>&1
run() { echo "== $*" >> $out; env "$@" SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 300 python tools/exp/exp_cob_timing.py 23 >> $out 2>&1; }
run SPLAHIP_COB_SHARE_PCT=90 SPLAHIP_COB_LEVELS=4
grep -v amdgpu.ids $out
