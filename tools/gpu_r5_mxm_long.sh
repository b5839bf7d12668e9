#!/bin/bash
# Round 5: from how many products a part of the block walk is taken by a wavefront of its own (SPLAHIP_MXM_LONG, compile time): side builds long<N>.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_mxm_long.txt
: > $out
for tag in "" long32 long512 long100000000 ""; do
  echo -n "lib tag '$tag': " >> $out
  SPLA_LIB_TAG=$tag timeout 300 python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms')" >> $out
done
