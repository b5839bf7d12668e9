#!/bin/bash
# round 4: first run of the persistent form: parity (mxv tests), then timing against the one-piece-per-workgroup launch
out=gpurun_out/r04_t2.txt
: > $out
timeout 900 python -m pytest tests/test_mxv_gpu.py tests/test_vxm_gpu.py tests/test_shard_gpu.py -x -q 2>&1 | tail -15 >> $out
for sc in 23 22; do
  timeout 300 python tools/exp/exp_r4_cob.py $sc 0 >> $out 2>&1
  timeout 300 python tools/exp/exp_r4_cob.py $sc 0 SPLAHIP_COB_PERSIST=0 >> $out 2>&1
done
timeout 300 python tools/exp/exp_r4_cob.py 23 1 >> $out 2>&1
cat $out
