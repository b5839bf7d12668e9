#!/bin/bash
# Round 5: the mxm knobs again after the call rework (split of hub rows, accumulator block, ranked form), R-MAT 18.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_mxm_knobs.txt
: > $out
run() { echo -n "$* : " >> $out; env "$@" timeout 300 python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms')" >> $out; }
run X=1
run SPLAHIP_MXM_SPLIT=131072
run SPLAHIP_MXM_SPLIT=262144
run SPLAHIP_MXM_SPLIT=1048576
run SPLAHIP_MXM_SPLIT=2097152
run SPLAHIP_MXM_BLOCK_KB=112
run SPLAHIP_MXM_BLOCK_KB=96
run SPLAHIP_MXM_RANKED=0
run SPLAHIP_MXM_LIGHT=2048
run X=2
