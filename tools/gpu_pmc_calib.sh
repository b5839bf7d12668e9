#!/bin/bash
# calibration of FETCH_SIZE on the sliced-ELL kernels: config B reads every entry once (522 MB of streams + ~50 MB of row data)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/pmc_calib"; mkdir -p "$R/gpurun_out"
for cfg in A B; do
SPLA_EXP_ONLY=$cfg timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_calib/$cfg" -- python3 "$R/tools/exp/exp_bits.py" x > "$R/gpurun_out/pmc_calib_$cfg.log" 2>&1; echo "pass $cfg exit $?"
done
cd "$R"; python - <<'PY'
import csv, glob
from collections import defaultdict
for f in sorted(glob.glob("gpurun_out/pmc_calib/**/*counter_collection.csv", recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = "BITS" if ("sell_mxv_kernel" in k and "true>" in k) else "VALUE" if "sell_mxv_kernel" in k else None
        if name: per[name][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for k, d in per.items():
        v = sorted(d.values()); print(f, k, "FETCH_SIZE KB median", v[len(v)//2], "n", len(v))
PY
find gpurun_out/pmc_calib -name "*.csv" -size +2M -delete
