#!/bin/bash
# Round 5: size of a light-rows batch (products / threads; side builds cap2048, cap2048t256) against the default 4096 / 512.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_mxm_batch.txt
: > $out
for tag in ${TAGS:-"" cap2048 cap2048t256 ""}; do
  echo -n "lib tag '$tag': " >> $out
  SPLA_LIB_TAG=$tag timeout 300 python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('verified'))" >> $out
done
