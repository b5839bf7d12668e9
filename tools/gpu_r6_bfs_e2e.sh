#!/bin/bash
# round 6: kernel trace of spla_Algorithm_bfs end to end (symmetrised R-MAT 20), per-kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r06b"
rm -rf "$R/gpurun_out/r06b/prof_bfs"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r06b/prof_bfs" --output-format csv -- python3 "$R/tools/exp/exp_bfs_e2e.py" 2>&1 | grep -v amdgpu.ids | tail -2
f=$(ls -t "$R"/gpurun_out/r06b/prof_bfs/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'calls':>6} {'avg us':>9} {'total ms':>9}  name   (21 bfs runs x 5 levels)")
for r in rows[:28]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.2f} {float(r['TotalDurationNs']) / 1e6:9.2f}  {r['Name'][:120]}")
PY
find "$R/gpurun_out/r06b/prof_bfs" -name "*.csv" -size +1M -delete
