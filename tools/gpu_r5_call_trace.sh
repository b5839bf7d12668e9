#!/bin/bash
# Round 5: kernel trace of ONE call of a bench workload: which kernels it launches, in order, with their durations and the GPU-idle gaps between them.
#   tools/gpu_r5_call_trace.sh <workload> [more bench args]      -> gpurun_out/r5_trace_<workload>.txt (the last 60 ms of the run)
w=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/calltrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/calltrace -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-verify "$@" > /dev/null 2>&1
f=$(find /tmp/calltrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/r5_trace_$w.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tend = int(rows[-1]["End_Timestamp"])
import os
rows = [r for r in rows if int(r["Start_Timestamp"]) > tend - float(os.environ.get("WINDOW_MS", "60")) * 1e6]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("splahip::", "").replace("(anonymous namespace)::", "")[:100]
    print(f"{(s - t0) / 1e6:9.3f} ms  gap {(s - prev_end) / 1e3:8.1f} us  dur {(e - s) / 1e3:9.1f} us  {name}")
    prev_end = max(prev_end, e)
P
