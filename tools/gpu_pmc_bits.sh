#!/bin/bash
# counters of the sliced-ELL BITS kernel on the BFS step (tools/exp/exp_bits.py, config A only); <= 3 counters per pass
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf "$R/gpurun_out/pmc_bits"; mkdir -p "$R/gpurun_out"
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\bSQ_[A-Z0-9_]+\b" | sort -u > "$R/gpurun_out/pmc_avail_sq.txt"; wc -l "$R/gpurun_out/pmc_avail_sq.txt"
pass() { n=$1; shift
  SPLA_EXP_ONLY=A timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_bits/$n" -- python3 "$R/tools/exp/exp_bits.py" x > "$R/gpurun_out/pmc_bits_$n.log" 2>&1; echo "pass $n exit $?"
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass c TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ_LATENCY
pass d SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pass e GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
cd "$R"; python - <<'PY'
import csv, glob
from collections import defaultdict
for f in sorted(glob.glob("gpurun_out/pmc_bits/**/*counter_collection.csv", recursive=True)):
    rows = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = "BITS" if ("sell_mxv_kernel" in k and "true>" in k) else "VALUE" if "sell_mxv_kernel" in k else None
        if name: rows[name][(row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
    for k, d in rows.items():
        per = defaultdict(list)
        for (c, disp), vals in d.items(): per[c].append(sum(vals))
        print(k, {c: round(sum(v) / len(v)) for c, v in per.items()}, "launches", len(next(iter(per.values()))))
PY
find gpurun_out/pmc_bits -name "*.csv" -size +2M -delete
