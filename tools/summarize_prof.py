#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: kernel_stats (per-kernel time) and counter_collection (per-kernel PMC averages),
keeping only this library's kernels (namespace splahip) unless --all."""
import csv
import sys
from collections import defaultdict
from pathlib import Path


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 150 else name[:147] + "..."


def main():
    root = Path(sys.argv[1])
    show_all = "--all" in sys.argv
    for f in sorted(root.rglob("*kernel_stats.csv")):
        print(f"== {f.relative_to(root)}")
        print(f"{'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
        for row in csv.DictReader(open(f)):
            if show_all or "splahip" in row["Name"]:
                print(f"{row['Calls']:>6} {float(row['AverageNs']) / 1e3:>10.2f} {float(row['MinNs']) / 1e3:>10.2f} {float(row['MaxNs']) / 1e3:>10.2f} "
                      f"{float(row['Percentage']):>6.2f}  {short(row['Name'])}")
    for f in sorted(root.rglob("*counter_collection.csv")):
        print(f"== {f.relative_to(root)}")
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            if show_all or "splahip" in name:
                acc[short(name)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, counters in acc.items():
            for c, vals in counters.items():
                print(f"{c:>18} n={len(vals):<4} avg={sum(vals) / len(vals):>16.1f} min={min(vals):>16.1f} max={max(vals):>16.1f}  {name}")


if __name__ == "__main__":
    main()
