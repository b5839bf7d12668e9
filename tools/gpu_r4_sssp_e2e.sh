#!/bin/bash
# round 4: kernel trace of spla_Algorithm_sssp end to end (symmetrised R-MAT 20, weights 1..255), per-kernel table per run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r04b"
rm -rf "$R/gpurun_out/r04b/prof_sssp"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r04b/prof_sssp" --output-format csv -- python3 "$R/tools/exp/exp_sssp_e2e.py" 2>&1 | grep "sssp e2e"
f=$(ls -t "$R"/gpurun_out/r04b/prof_sssp/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
print(f"{'calls':>6} {'avg us':>9} {'us per run':>11}  name   (11 sssp runs)")
for r in rows:
    n = r['Name']
    c = int(r['Calls'])
    if c < 11 or 'at::' in n or 'Histogram' in n: continue
    t = float(r['TotalDurationNs']) / 1e3 / 11
    tot += t
    if t > 8: print(f"{c:6d} {float(r['AverageNs']) / 1e3:9.2f} {t:11.1f}  {n[:110]}")
print('kernel time per run (us):', round(tot, 1))
PY
find "$R/gpurun_out/r04b/prof_sssp" -name "*.csv" -size +1M -delete
