#!/bin/bash
# round 5: the triangle-count form of mxmT (owner form on / off): tests, time per call, kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp; cd "$R"; mkdir -p gpurun_out
timeout 300 python3 -m pytest tests/test_mxm_gpu.py -x -q -m gpu --timeout 300 2>&1 | tail -2
for o in 1 0; do
  SPLAHIP_MXMT_OWNER=$o timeout 300 python3 bench.py --workload tc-mxmT --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('owner form $o:', d['ms_per_step'], 'ms', d['config']['triangles'], d.get('verified'))"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tc -- python3 bench.py --workload tc-mxmT --steps 10 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
f=$(find gpurun_out/prof_tc -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    if "at::" in r["Name"]: continue
    print(f'  {r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
find gpurun_out/prof_tc -name "*kernel_trace.csv" -delete
