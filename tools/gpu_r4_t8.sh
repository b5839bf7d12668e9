#!/bin/bash
out=gpurun_out/r04_t8.txt
: > $out
run() { echo "== $*" >> $out; env "$@" SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 300 python tools/exp/exp_cob_timing.py 23 2>&1 | grep -v "amdgpu.ids\|schedule 0" >> $out; }
run SPLAHIP_COB_GSS_K10=10
cat $out
