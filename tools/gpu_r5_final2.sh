#!/bin/bash
# Round 5, the last build: GPU suite and the default bench line (what profiles/r05_bench.json / r05_pytest_gpu.txt hold), kernel statistics of the contract-line command.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu --timeout 900 > gpurun_out/r5_pytest_gpu.txt 2>&1
grep "passed\|failed" gpurun_out/r5_pytest_gpu.txt
timeout 900 python3 bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r5_bench.json'))
print(d['value'], d['unit'], d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['verified'], d['cpu_baseline']['value'])
for v in d['variants']:
    if 'value' in v: print('  ', v.get('name','')[:60], v['value'], v['unit'], v['ms_per_step'], v.get('verified'))
"
cd /tmp
rm -rf "$R/gpurun_out/prof_r05"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r05" --output-format csv -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-verify > "$R/gpurun_out/r05_bench_under_rocprof.json" 2> "$R/gpurun_out/r05_bench_under_rocprof.err"
f=$(ls "$R"/gpurun_out/prof_r05/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" > "$R/gpurun_out/r05_rocprof_kernel_stats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify   (kernel names cut to 150 characters)")
print(f"{'calls':>6} {'avg us':>10} {'total ms':>10} {'%':>6}  name")
for r in rows[:40]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}  {r['Name'][:150]}")
PY
t=$(ls "$R"/gpurun_out/prof_r05/*/*kernel_trace.csv 2>/dev/null | tail -1); python3 "$R/tools/cob_trace_groups.py" "$t" >> "$R/gpurun_out/r05_rocprof_kernel_stats.txt"
find "$R/gpurun_out/prof_r05" -name "*.csv" -size +1M -delete
tail -8 "$R/gpurun_out/r05_rocprof_kernel_stats.txt" | cut -c1-160
