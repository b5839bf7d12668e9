#!/bin/bash
out=gpurun_out/r04_t12.txt
: > $out
timeout 900 python -m pytest tests/test_mxv_gpu.py -x -q 2>&1 | tail -40 >> $out
timeout 600 python tools/exp/exp_r4_mask.py 23 2>&1 | grep -v amdgpu.ids >> $out
cat $out
