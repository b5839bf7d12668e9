#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/gpu_round.sh (FETCH_SIZE, WRITE_SIZE; one counter per pass) into the
per-launch HBM-side traffic of bench.py's timed kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for gfx950: FETCH_SIZE is reported in KB and counts 128-B requests as 64 B (x2; re-calibrated on this kernel's own
access pattern, profiles/r01_exp_gather.txt: known 557 MB -> 293 MB reported), WRITE_SIZE is exact.
    python tools/pmc_traffic.py gpurun_out profiles/pmc_traffic.json '<kernel substring>' <bench.json>
The timed configuration's dispatches are the most frequent value cluster of that kernel (warm-up + steps)."""
import csv
import glob
import json
import statistics
import sys
from collections import Counter


def cluster_median(vals):
    # dispatches of the timed configuration dominate: take the median of the values within 2 % of the mode bucket
    buckets = Counter(round(v / (0.02 * max(vals))) for v in vals)
    b = buckets.most_common(1)[0][0]
    sel = [v for v in vals if round(v / (0.02 * max(vals))) == b]
    return statistics.median(sel), len(sel)


def main():
    root, out, needle, bench = sys.argv[1:5]
    res = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        import os
        f = max(glob.glob(f"{root}/pmc_{kind}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        rows = list(csv.DictReader(open(f)))
        total, disp = 0.0, []
        for nd in needle.split("|"):  # a step made of several launches: the sum of every kernel's per-launch figure
            # one row per hardware instance: sum the instances of a dispatch first
            per = {}
            for r in rows:
                if nd in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if not per:
                continue
            m, n = cluster_median(list(per.values())) if max(per.values()) > 0 else (0.0, len(per))
            total += m
            disp.append(n)
        res[counter], res[counter + "_dispatches"] = total, disp
    b = json.loads(open(bench).read().strip().splitlines()[-1])
    traffic = (2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0
    doc = {"kernel": needle, "fetch_size_kb_reported": res["FETCH_SIZE"], "write_size_kb_reported": res["WRITE_SIZE"],
           "dispatches": [res["FETCH_SIZE_dispatches"], res["WRITE_SIZE_dispatches"]],
           "traffic_bytes": round(traffic), "algorithmic_bytes": b["roofline"]["algorithmic_bytes"],
           "traffic_over_algorithmic": round(traffic / b["roofline"]["algorithmic_bytes"], 3),
           "config": {k: b["config"][k] for k in ("n", "nnz", "nnz_selected")},
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "
                     "--no-verify --no-variants`; traffic = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950 half-count of 128-B read requests; the counter sits on "
                     "the L2's fabric side, so Infinity-Cache hits — here the gathered vector's lines — are included)"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
