#!/bin/bash
# round 5: mxm A x A at R-MAT 18: time per call, phases (SPLAHIP_MXM_TRACE), kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp; cd "$R"; mkdir -p gpurun_out
timeout 300 python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/mxm.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mxm:', d['ms_per_step'], 'ms', d['value'], d['unit'], d.get('verified'), d['config'].get('row_classes'))"
SPLAHIP_MXM_TRACE=1 timeout 300 python3 bench.py --workload mxm --steps 1 --warmup 1 --no-cpu-baseline --no-verify 2>&1 >/dev/null | grep "\[mxm" | tail -24
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mxm -- python3 bench.py --workload mxm --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
f=$(find gpurun_out/prof_mxm -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
n=0
for r in rows:
    if "at::" in r["Name"]: continue
    print(f'  {r["Name"][:110]:110s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
    n+=1
    if n>=16: break
PY
find gpurun_out/prof_mxm -name "*kernel_trace.csv" -delete
