#!/bin/bash
# Round 5: the host side of an mxm call (pinned copies, stream-ordered scratch, host work under the light rows' kernel): host marks, the workload line, a kernel trace, parity tests.
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5_mxm_host.txt
: > $out
SPLAHIP_MXM_HOSTTRACE=1 timeout 600 python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline 2> /tmp/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', 'verified', d.get('verified'))" >> $out
grep "mxm-host" /tmp/err.txt | tail -26 >> $out
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "mxm" --timeout 900 2>&1 | tail -3 >> $out
bash tools/gpu_r5_mxm_trace.sh
