#!/bin/bash
# Round 5: the four-deep product walk of the blocked mxm kernels: phase ticks before (lib tag spatime) / after (spatime2), the mxm workload line, the mxm parity tests.
export TMPDIR=/tmp
out=gpurun_out/r5_mxm_walk.txt
: > $out
for tag in spatime spatime2; do
  echo "== $tag" >> $out
  SPLA_LIB_TAG=$tag timeout 300 python3 tools/exp/exp_spa_timing.py 18 2>&1 | tail -2 >> $out
done
echo "== mxm workload (main lib)" >> $out
SPLAHIP_MXM_TRACE=1 timeout 600 python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline 2>> $out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', 'verified', d.get('verified'))" >> $out
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "mxm" --timeout 900 2>&1 | tail -3 >> $out
