#!/bin/bash
# Round 5: kernel trace of the headline loop (bench.py --no-variants): the kernels of consecutive calls and the gaps between them.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/headtrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/headtrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify --no-variants > /dev/null 2>&1
f=$(find /tmp/headtrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/r5_trace_headline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cob_mxv_kernel" in r["Kernel_Name"]]
rows = rows[idx[-8] - 2:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("splahip::", "").replace("(anonymous namespace)::", "")[:80]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  dur {(e - s) / 1e3:8.1f} us  {name}")
    prev_end = max(prev_end, e)
P
