#!/bin/bash
# round 4: compile-time variants of vstat_kernel (SPLA_LIB_TAG builds) — kernel times from a trace of the BFS-step line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r04b"
for tag in "" ${TAGS:-vsb128 vsb512 vsnos}; do
  export SPLA_LIB_TAG=$tag
  rm -rf "$R/gpurun_out/r04b/prof_vs"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r04b/prof_vs" --output-format csv -- python3 "$R/bench.py" --workload bfs-step --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-variants > /dev/null 2>&1
  f=$(ls -t "$R"/gpurun_out/r04b/prof_vs/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'vstat' in r['Name'] or 'first_hit_kernel' in r['Name']:
        print(f"tag={sys.argv[2]:8s} {float(r['AverageNs']) / 1e3:8.2f} us  {r['Name'][:60]}")
PY
  find "$R/gpurun_out/r04b/prof_vs" -name "*.csv" -size +1M -delete
done
