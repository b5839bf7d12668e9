#!/bin/bash
# round 4: timeline of the persistent form under schedule knobs
out=gpurun_out/r04_t3.txt
: > $out
run() { echo "== $*" >> $out; env "$@" SPLAHIP_COB_TRACE=1 SPLA_LIB_TAG=cobtime timeout 300 python tools/exp/exp_cob_timing.py 23 >> $out 2>&1; }
run SPLAHIP_COB_SHARE_PCT=90
run SPLAHIP_COB_SHARE_PCT=90 SPLAHIP_COB_LEVELS=4
run SPLAHIP_COB_SHARE_PCT=0
cat $out
