#!/bin/bash
out=gpurun_out/r04_t9.txt
: > $out
timeout 900 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $out
for cfg in "10 128 0" "15 128 0" "7 128 0" "10 256 0"; do
  set -- $cfg
  timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_GSS_K10=$1 SPLAHIP_COB_GSS_MIN=$2 SPLAHIP_COB_GSS_ORDER=$3 EXP_CHECK=$([ "$cfg" = "10 128 0" ] && echo 1 || echo 0) 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $out
done
timeout 300 python tools/exp/exp_r4_cob.py 23 0 SPLAHIP_COB_GSS=0 SPLAHIP_COB_SHARE_PCT=90 SPLAHIP_COB_LEVELS=3 EXP_CHECK=0 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $out
timeout 300 python tools/exp/exp_r4_cob.py 22 0 SPLAHIP_COB_GSS_K10=10 EXP_CHECK=0 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $out
timeout 300 python tools/exp/exp_r4_cob.py 22 0 SPLAHIP_COB_GSS=0 SPLAHIP_COB_SHARE_PCT=110 SPLAHIP_COB_LEVELS=3 EXP_CHECK=0 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $out
run() { echo "== $*" >> $out; env "$@" SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 300 python tools/exp/exp_cob_timing.py 23 2>&1 | grep -v "amdgpu.ids\|schedule 0" >> $out; }
run SPLAHIP_COB_GSS_K10=10
cat $out
