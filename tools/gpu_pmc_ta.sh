#!/bin/bash
# texture-addresser / vector-L1 counters of the sliced-ELL kernel vs the CSR wave-tile kernel (two --pmc passes, no traces besides kernel-trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/pmc_ta"; mkdir -p "$R/gpurun_out"
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA|TCP|TD|TCC)_[A-Z0-9_]+\b" | sort -u > "$R/gpurun_out/pmc_avail_ta_tcp.txt"; wc -l "$R/gpurun_out/pmc_avail_ta_tcp.txt"
pass() { # name counters...
  n=$1; shift
  timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_ta/$n" -- python3 "$R/tools/exp/exp_sell_min.py" x > "$R/gpurun_out/pmc_ta_$n.log" 2>&1; echo "pass $n exit $?"
}
pass a TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE
pass b TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
pass c TA_DATA_STALLED_BY_TC_CYCLES TCP_TCC_READ_REQ_LATENCY
cd "$R"; python - <<'PY'
import csv, glob
from collections import defaultdict
for f in sorted(glob.glob("gpurun_out/pmc_ta/**/*counter_collection.csv", recursive=True)):
    rows = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = "sell_mxv" if "sell_mxv_kernel" in k else "wavetile" if "spmv_wavetile" in k else None
        if name: rows[name][(row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
    for k, d in rows.items():
        per = defaultdict(list)
        for (c, disp), vals in d.items(): per[c].append((sum(vals), len(vals)))   # sum over hardware instances of one dispatch
        print(k, {c: (round(sum(x for x, _ in v) / len(v)), "instances", v[0][1]) for c, v in per.items()}, "launches", len(next(iter(per.values()))))
PY
find gpurun_out/pmc_ta -name "*.csv" -size +2M -delete
