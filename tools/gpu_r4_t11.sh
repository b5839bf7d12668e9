#!/bin/bash
out=gpurun_out/r04_t11.txt
: > $out
timeout 600 python tools/exp/exp_r4_mask.py 23 SPLAHIP_COB=0 2>&1 | grep -v amdgpu.ids >> $out
timeout 600 python tools/exp/exp_r4_mask.py 22 SPLAHIP_COB=0 2>&1 | grep -v amdgpu.ids >> $out
cat $out
