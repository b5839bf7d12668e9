#!/bin/bash
# round 6: kernel trace of spla_Algorithm_sssp / pr end to end (symmetrised R-MAT 20): the timeline of the last run (start, gap to the previous kernel, duration)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r06b"
for which in sssp pr; do
  rm -rf "$R/gpurun_out/r06b/prof_$which"
  timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/r06b/prof_$which" --output-format csv -- python3 "$R/tools/exp/exp_r6_algo_trace.py" $which 2>&1 | grep "e2e ms"
  f=$(ls -t "$R"/gpurun_out/r06b/prof_$which/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 - "$f" > "$R/gpurun_out/r6_trace_$which.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-400:]
t0 = int(rows[0]['Start_Timestamp']); prev = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:110]}")
    prev = e
PY
  find "$R/gpurun_out/r06b/prof_$which" -name "*.csv" -size +1M -delete
done
