#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `bench.py --no-variants`, the duration of the bench's TIMED launches of the dominant
kernel: the last K launches of that kernel are the timed region (K = --steps); compare with the HIP-event figure bench.py printed.
    python tools/timed_kernel_from_trace.py <prof dir> '<kernel substring>' <steps> <bench json>"""
import csv
import glob
import json
import statistics
import sys

root, needle, steps, bench = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
f = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getmtime(p))
rows = [r for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-steps:]
b = json.loads(open(bench).read().strip().splitlines()[-1])
print(f"kernel: {needle}")
print(f"launches of this kernel in the trace: {len(rows)}; timed region = last {len(dur)}")
print(f"rocprofv3 kernel-trace: avg {statistics.mean(dur):.2f} us, median {statistics.median(dur):.2f} us, min {min(dur):.2f}, max {max(dur):.2f}")
print(f"bench.py HIP events (same command, same run): avg {b['roofline']['kernel_avg_us']} us, median {b['roofline']['kernel_median_us']} us")
