#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes: for every kernel whose name matches one of the given substrings, the median over dispatches of
(counter summed over hardware instances).  FETCH_SIZE / WRITE_SIZE are reported in KB; HBM-side traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(gfx950 counts a 128-B read request as 64 B: /opt/skills/guides/MI355X_MICROARCH.md, HBM section).
    python tools/pmc_kernels.py <dir with pass sub-directories> <substr> [<substr> ...]"""
import csv
import glob
import json
import statistics
import sys
from collections import defaultdict


def main():
    root, needles = sys.argv[1], sys.argv[2:]
    out = defaultdict(dict)
    for f in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
        per = defaultdict(lambda: defaultdict(float))  # (needle, counter) -> dispatch -> sum over instances
        for r in csv.DictReader(open(f)):
            for nd in needles:
                if nd in r["Kernel_Name"]:
                    per[(nd, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (nd, c), d in per.items():
            vals = sorted(d.values())
            # a bench runs several configurations through one kernel: report the most common size class (values within 3 % of the median of the upper half)
            out[nd][c] = {"median": statistics.median(vals), "max": vals[-1], "dispatches": len(vals)}
    for nd, cs in out.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            cs["traffic_bytes_median"] = round((2 * cs["FETCH_SIZE"]["median"] + cs["WRITE_SIZE"]["median"]) * 1024)
            cs["traffic_bytes_max"] = round((2 * cs["FETCH_SIZE"]["max"] + cs["WRITE_SIZE"]["max"]) * 1024)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
