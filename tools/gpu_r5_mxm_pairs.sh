#!/bin/bash
# Round 5: (column, value) pairs of B in the numeric walks of mxm: parity tests, the workload line, the kernels of a call.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_mxm_pairs.txt
: > $out
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "mxm" --timeout 900 2>&1 | grep "passed\|failed" >> $out
for i in 1 2; do timeout 600 python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', 'verified', d.get('verified'))" >> $out; done
timeout 300 python3 tools/exp/fuzz_ops.py 60 777 2>&1 | tail -1 >> $out
bash tools/gpu_r5_mxm_trace.sh
grep "mxm_spa\|light_batch\|pairs" $GRAFT_REPO_ROOT/gpurun_out/r5_mxm_trace.txt | cut -c1-150 >> $out
