#!/bin/bash
# round 4: kernel times of the BFS-step line (first-hit form) from a kernel trace, for SPLAHIP_FIRST_HIT=1 / 0
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r04b"
for fh in 1 0; do
  export SPLAHIP_FIRST_HIT=$fh
  rm -rf "$R/gpurun_out/r04b/prof_fh$fh"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r04b/prof_fh$fh" --output-format csv -- python3 "$R/bench.py" --workload bfs-step --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-variants > "$R/gpurun_out/r04b/bfs_fh$fh.json" 2> /dev/null
  f=$(ls -t "$R"/gpurun_out/r04b/prof_fh$fh/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== SPLAHIP_FIRST_HIT=$fh"; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r04b/bfs_fh$fh.json')); print(d['ms_per_step'], d['roofline']['kernel'][:60], d['roofline']['kernel_avg_us'])"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('vstat', 'first_hit', 'sell_', 'rowseq')):
        print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} us  {r['Name'][:110]}")
PY
  find "$R/gpurun_out/r04b/prof_fh$fh" -name "*.csv" -size +1M -delete
done
