#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of `bench.py --no-variants`, the duration of the bench's TIMED step: the step is a fixed
sequence of launches (needles separated by '|', e.g. the sliced-ELL call = vprep + main kernel + gated value-form kernel + combine +
cold list); the last K launches of each are the timed region (K = --steps).  Prints every kernel's average and their sum, next to
the HIP-event figure bench.py printed for the whole call (events additionally contain the gaps between the launches).
    python tools/timed_kernel_from_trace.py <prof dir> '<kernel substring>[|<kernel substring>...]' <steps> <bench json>"""
import csv
import glob
import json
import os
import statistics
import sys

root, needles, steps, bench = sys.argv[1], sys.argv[2].split("|"), int(sys.argv[3]), sys.argv[4]
f = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
allrows = list(csv.DictReader(open(f)))
b = json.loads(open(bench).read().strip().splitlines()[-1])
total_avg, total_med = 0.0, 0.0
for needle in needles:
    rows = [r for r in allrows if needle in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-steps:]
    if not dur:
        print(f"kernel: {needle}: not in the trace")
        continue
    total_avg += statistics.mean(dur)
    total_med += statistics.median(dur)
    print(f"kernel: {needle}")
    print(f"  launches in the trace: {len(rows)}; timed region = last {len(dur)}: avg {statistics.mean(dur):.2f} us, median {statistics.median(dur):.2f} us, "
          f"min {min(dur):.2f}, max {max(dur):.2f}")
print(f"sum over the step's kernels (rocprofv3 kernel-trace): avg {total_avg:.2f} us, median {total_med:.2f} us")
print(f"bench.py HIP events around the whole call (same command, same run; includes the gaps between its launches): "
      f"avg {b['roofline']['kernel_avg_us']} us, median {b['roofline']['kernel_median_us']} us")
