#!/bin/bash
# Round 5: kernel trace of the mxm workload (which kernels a call launches, in order, with their durations and the gaps between them).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mxmtrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/mxmtrace -- python3 $GRAFT_REPO_ROOT/bench.py --workload mxm --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
f=$(find /tmp/mxmtrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/r5_mxm_trace.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last rowprod_kernel on
idx = [i for i, r in enumerate(rows) if "rowprod_kernel" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("splahip::", "").replace("(anonymous namespace)::", "")[:90]
    print(f"{(s - t0) / 1e6:9.3f} ms  gap {(s - prev_end) / 1e3:8.1f} us  dur {(e - s) / 1e3:9.1f} us  grid {r.get('Grid_Size', '?'):>9}  {name}")
    prev_end = max(prev_end, e)
    if (s - t0) > 80e6: break
P
