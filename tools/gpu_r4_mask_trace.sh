#!/bin/bash
# round 4: kernel times of the masked calls (tools/exp/exp_r4_mask.py) from a kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r04b"
rm -rf "$R/gpurun_out/r04b/prof_mask"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r04b/prof_mask" --output-format csv -- python3 "$R/tools/exp/exp_r4_mask.py" 23 2>&1 | grep -E "R-MAT|%"
f=$(ls -t "$R"/gpurun_out/r04b/prof_mask/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('rows_', 'probe', 'wavetile', 'combine', 'cob_')) and 'at::' not in n:
        print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.2f} us  {n[:100]}")
PY
find "$R/gpurun_out/r04b/prof_mask" -name "*.csv" -size +1M -delete
