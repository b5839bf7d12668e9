#!/bin/bash
# round 5: kernel trace of the fp32 plus-times call under the static pieces and under the run-time schedule
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp; cd "$R"; mkdir -p gpurun_out
for d in 0 1; do
  SPLAHIP_COB_DYN=$d timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dyn$d -- python3 tools/exp/exp_r5_dyn_once.py ${1:-23} > gpurun_out/prof_dyn$d.log 2>&1
  grep "scale=" gpurun_out/prof_dyn$d.log
  f=$(find gpurun_out/prof_dyn$d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f'  {r["Name"][:110]:110s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.2f} us  min {float(r["MinNs"])/1e3:9.2f}  max {float(r["MaxNs"])/1e3:9.2f}')
PY
  find gpurun_out/prof_dyn$d -name "*kernel_trace.csv" -delete
done
