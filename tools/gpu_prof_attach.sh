#!/bin/bash
# kernel trace of one sliced-ELL plan build (R-MAT scale 22)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/prof_attach"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_attach" -- python3 "$R/tools/exp/exp_sell_min.py" x > /dev/null 2>&1; echo "exit $?"
cd "$R"; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_attach/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sell_" in r["Name"] and "sell_mxv" not in r["Name"] or "rocprim" in r["Name"].lower()]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]: print(f"{int(r['Calls']):4d} calls {float(r['TotalDurationNs'])/1e6:8.2f} ms total  {r['Name'][:110]}")
PY
rm -rf gpurun_out/prof_attach
