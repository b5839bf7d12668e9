#!/bin/bash
# Round 6, final build: smoke(), then kernel trace + stats of the contract-line command (the last part of tools/gpu_pmc_r06.sh on its own)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
cd /tmp
rm -rf "$R/gpurun_out/prof_r06"
timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r06" --output-format csv -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-verify > "$R/gpurun_out/r06_bench_under_rocprof.json" 2> "$R/gpurun_out/r06_bench_under_rocprof.err"
f=$(ls "$R"/gpurun_out/prof_r06/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== rocprof kernel stats ($f)"
python3 - "$f" <<'PY' | tee "$R/gpurun_out/r06_rocprof_kernel_stats.txt" | grep "cob_\|rows_\|first_hit" | head -12
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify   (kernel names cut to 150 characters)")
print(f"{'calls':>6} {'avg us':>10} {'total ms':>10} {'%':>6}  name")
for r in rows[:48]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}  {r['Name'][:150]}")
PY
t=$(ls "$R"/gpurun_out/prof_r06/*/*kernel_trace.csv 2>/dev/null | tail -1); python3 "$R/tools/cob_trace_groups.py" "$t" | tee -a "$R/gpurun_out/r06_rocprof_kernel_stats.txt" | tail -6
find "$R/gpurun_out/prof_r06" -name "*.csv" -size +1M -delete
python3 -c "
import json; d=json.loads(open('$R/gpurun_out/r06_bench_under_rocprof.json').read().strip().splitlines()[-1]); print('under rocprof:', d['ms_per_step'], d['roofline']['kernel_avg_us'], d['roofline']['frac'])"
